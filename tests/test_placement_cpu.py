"""CPU placement (device.cpu_plan) and the failure paths of the batch threads: host logic, no GPU."""
import os
import threading

import pytest

from solex_ser_recon_en_amd import device as dev


def _fake_sysfs(tmp_path, node, cpus_per_group, n_groups, smt_offset):
    """A node of n_groups L3 groups, each cpus_per_group cores + their SMT siblings at +smt_offset."""
    all_cpus = []
    for g in range(n_groups):
        cores = list(range(g * cpus_per_group, (g + 1) * cpus_per_group))
        grp = cores + [c + smt_offset for c in cores]
        text = '%d-%d,%d-%d' % (cores[0], cores[-1], cores[0] + smt_offset, cores[-1] + smt_offset)
        for c in grp:
            d = tmp_path / 'devices/system/cpu' / ('cpu%d' % c) / 'cache/index3'
            d.mkdir(parents=True)
            (d / 'shared_cpu_list').write_text(text + '\n')
        all_cpus += grp
    d = tmp_path / 'devices/system/node' / ('node%d' % node)
    d.mkdir(parents=True)
    (d / 'cpulist').write_text('0-%d,%d-%d\n' % (n_groups * cpus_per_group - 1, smt_offset, smt_offset + n_groups * cpus_per_group - 1))
    return set(all_cpus)


def test_cpulist_parses_ranges_singles_and_blanks():
    assert dev._cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert dev._cpulist('') == []
    assert dev._cpulist('5') == [5]


def test_l3_groups_follow_the_cache_topology_and_skip_offline_cpus(tmp_path):
    online = _fake_sysfs(tmp_path, 0, 8, 4, 128)
    groups = dev._l3_groups(0, online, sysfs=str(tmp_path))
    assert len(groups) == 4
    assert groups[0] == set(range(0, 8)) | set(range(128, 136))
    assert groups[3] == set(range(24, 32)) | set(range(152, 160))
    fewer = online - {3, 131} - set(range(8, 16)) - set(range(136, 144))          # one core off, one whole group off
    groups = dev._l3_groups(0, fewer, sysfs=str(tmp_path))
    assert len(groups) == 3 and groups[0] == (set(range(0, 8)) | set(range(128, 136))) - {3, 131}


def test_gpus_on_one_node_get_disjoint_shares():
    groups = [set(range(8 * g, 8 * g + 8)) for g in range(8)]
    shares = [dev._share_of_node(groups, k, 4) for k in range(4)]
    for k, s in enumerate(shares):
        assert s['scan'] == groups[k] and s['io'] == groups[k + 4]
    used = [c for s in shares for c in s['scan'] | s['io']]
    assert len(used) == len(set(used)) == 64
    alone = dev._share_of_node(groups[:1], 0, 1)                                   # one group: readers share it
    assert alone['scan'] == alone['io'] == groups[0]
    assert dev._share_of_node(groups[:2], 3, 4) is None                            # more GPUs than groups: no share
    assert dev._share_of_node([{0, 1}], 0, 1) is None                              # too small to pin four workers to


def test_plan_stays_inside_the_cpus_the_process_may_use():
    """Two sockets of 4 L3 groups x 8 cpus, GPUs 0-3 on node 0 and 4-7 on node 1."""
    node_groups = {0: [set(range(8 * g, 8 * g + 8)) for g in range(4)], 1: [set(range(32 + 8 * g, 40 + 8 * g)) for g in range(4)]}
    everything = set(range(64))
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    pci = [(0, 10 * i, 0) for i in range(8)]

    def of_node(n):
        return node_groups[n]

    def of_cpus(cpus):
        return [g for n in (0, 1) for g in node_groups[n] if g & set(cpus)]
    # the whole machine: GPU k of a node gets that node's k-th group
    for i in range(8):
        plan = dev._plan_within(everything, nodes, pci, i, of_node, of_cpus)
        assert plan['scan'] == node_groups[nodes[i]][i % 4] and plan['io'] == plan['scan']
    # a cpuset that only holds node 0's first two groups: GPUs 0 and 1 keep a group each, 2 and 3 get none of their own
    half = set(range(16))
    assert dev._plan_within(half, nodes, pci, 0, of_node, of_cpus)['scan'] == set(range(8))
    assert dev._plan_within(half, nodes, pci, 1, of_node, of_cpus)['scan'] == set(range(8, 16))
    assert dev._plan_within(half, nodes, pci, 2, of_node, of_cpus) is None
    # GPU 5 (node 1) confined to node 0's cpus: a group of what is allowed, never a cpu outside it
    plan = dev._plan_within(set(range(32)), nodes, pci, 5, of_node, of_cpus)
    assert plan['scan'] == set(range(8, 16)) and plan['io'] <= set(range(32))
    # a mask of one group, or of a few cores: left alone
    assert dev._plan_within(set(range(40, 48)), nodes, pci, 5, of_node, of_cpus) is None
    assert dev._plan_within(set(range(40, 48)), nodes, pci, 4, of_node, of_cpus) == {'scan': set(range(40, 48)), 'io': set(range(40, 48))}
    assert dev._plan_within({3, 4}, nodes, pci, 0, of_node, of_cpus) is None
    assert dev._plan_within(set(range(8)), [-1], [(0, 0, 0)], 0, of_node, of_cpus) is None      # unknown node, one group
    # every rank sees only its own GPU (index 0 everywhere): the ranks of a node take its groups by local rank
    seen = [dev._plan_within(everything, [0], [(0, 0, 0)], 0, of_node, of_cpus, local_rank=r)['scan'] for r in range(4)]
    assert seen == node_groups[0]


def test_scan_workers_go_to_the_quietest_group_of_the_gpus_share(tmp_path):
    groups = [set(range(8 * g, 8 * g + 8)) for g in range(8)]
    busy = {c: 0.0 for c in range(64)}
    busy.update({c: 0.9 for c in range(0, 8)})                                     # another tenant sits on the first group
    plan = dev._share_of_node(groups, 0, 1, busy)
    assert plan['scan'] == groups[1] and plan['io'] == set(range(64)) - groups[1]
    assert dev._share_of_node(groups, 0, 4, busy)['scan'] == groups[4]             # this GPU's share: groups 0 and 4
    assert dev._share_of_node(groups, 1, 4, busy)['scan'] == groups[1]             # untouched shares keep their order
    assert dev._share_of_node(groups, 0, 1, None)['scan'] == groups[0]
    stat = tmp_path / 'stat'
    stat.write_text('cpu  10 0 10 100 0 0 0 0 0 0\ncpu0 5 0 5 50 0 0 0 0 0 0\ncpu1 5 0 5 50 0 0 0 0 0 0\nintr 1\n')
    assert dev._cpu_busy(0.0, str(stat)) == {0: 0.0, 1: 0.0}


def test_cpu_plan_env_override_and_off(monkeypatch):
    import torch
    monkeypatch.setattr(dev, '_cpu_plan', {})
    monkeypatch.setenv('SHG_CPU_AFFINITY', 'off')
    assert dev.cpu_plan(torch.device('cuda', 0)) is None
    monkeypatch.setattr(dev, '_cpu_plan', {})
    monkeypatch.setenv('SHG_CPU_AFFINITY', '2-3,6')
    assert dev.cpu_plan(torch.device('cuda', 0)) == {'scan': {2, 3, 6}, 'io': {2, 3, 6}}
    assert dev.cpu_plan(torch.device('cuda', 0)) is dev._cpu_plan[0]               # cached per device


def test_cpu_plan_never_raises_when_the_topology_cannot_be_read(monkeypatch):
    import torch
    monkeypatch.setattr(dev, '_cpu_plan', {})
    monkeypatch.setenv('SHG_CPU_AFFINITY', 'auto')
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(dev._cpulist(open('/sys/devices/system/cpu/online').read())))

    def boom(*a, **k):
        raise AssertionError('Invalid device id')                                  # what torch raises without a GPU
    monkeypatch.setattr(torch.cuda, 'get_device_properties', boom)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    assert dev.cpu_plan(torch.device('cuda', 0)) is None
    assert dev.bind_thread('scan', torch.device('cuda', 0)) is None


def test_bind_thread_sets_and_returns_the_previous_mask(monkeypatch):
    import torch
    mine = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(dev, '_cpu_plan', {0: {'scan': {mine[0]}, 'io': {mine[-1]}}})
    seen = {}

    def body():
        seen['old'] = dev.bind_thread('scan', torch.device('cuda', 0))
        seen['now'] = os.sched_getaffinity(0)
    t = threading.Thread(target=body)
    t.start()
    t.join()
    assert seen['old'] == set(mine) and seen['now'] == {mine[0]}
    assert os.sched_getaffinity(0) == set(mine)                                    # per thread: the caller is untouched


def test_a_decoder_thread_that_cannot_start_fails_its_files_instead_of_hanging(monkeypatch):
    import torch
    from solex_ser_recon_en_amd import Solex_recon as sr
    monkeypatch.setattr(sr, 'default_device', lambda: torch.device('cuda', 0))

    def no_gpu(device):
        raise RuntimeError('no such device')
    monkeypatch.setattr(torch.cuda, 'set_device', no_gpu)
    d = sr._Decoder([('a.ser', {}), ('b.ser', {})], None, ahead=2)
    for i in range(2):
        with pytest.raises(RuntimeError, match='no such device'):
            d.get(i)
    d.cancel()
    assert d.finished.is_set()


def test_service_threads_survive_a_failing_job_and_are_reused():
    from solex_ser_recon_en_amd import Solex_recon as sr
    a = sr._Service.named('shg-test-service')
    seen = []
    done = threading.Event()

    def bad():
        raise RuntimeError('job failed')
    a.jobs.put(bad)
    a.jobs.put(lambda: (seen.append(threading.current_thread().name), done.set()))
    assert done.wait(10) and seen == ['shg-test-service']
    assert sr._Service.named('shg-test-service') is a and a.thread.daemon


def test_a_worker_error_outside_a_scan_fails_the_batch(monkeypatch):
    """An exception a scan worker meets outside scan(i) -- here its final stream.synchronize() -- is raised by the batch, not
    swallowed by the service thread."""
    import contextlib
    import torch
    from solex_ser_recon_en_amd import Solex_recon as sr

    class FakeStream:
        def synchronize(self):
            raise RuntimeError('device lost')
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'stream', lambda s: contextlib.nullcontext())
    monkeypatch.setattr(sr, '_worker_context', lambda device, k: {'stream': FakeStream(), 'buffers': {}})
    monkeypatch.setattr(sr, 'bind_thread', lambda kind, device=None: None)
    done = []
    with pytest.raises(RuntimeError, match='device lost'):
        sr._scan_pool(done.append, 3, 2, torch.device('cuda', 0))
    assert sorted(done) == [0, 1, 2]
