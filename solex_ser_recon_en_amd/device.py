"""Device-resident images behind a NumPy-looking handle.

The reference passes NumPy arrays between its stages (disk_list, frame_circularized,
cc, ...).  Here those objects stay in HBM between stages; DeviceImage gives callers the
ndarray surface they expect (shape, dtype, indexing, np.asarray) by copying to the host
lazily and once.
"""
import numpy as np
import torch


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible: the SHG hot path runs on MI355X only (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


_cpu_plan = {}


def _cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if part:
            a, _, b = part.partition('-')
            cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _l3_groups(node, online, sysfs='/sys', cpus=None):
    """The L3 groups (sets of online cpus that share one last-level cache) of a NUMA node -- or of the given cpus --
    in cpu order."""
    groups, seen = [], set()
    if cpus is None:
        cpus = _cpulist(open('%s/devices/system/node/node%d/cpulist' % (sysfs, node)).read())
    for c in sorted(cpus):
        if c in seen or c not in online:
            continue
        grp = set(_cpulist(open('%s/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list' % (sysfs, c)).read())) & online
        seen |= grp | {c}
        if grp:
            groups.append(grp)
    return groups


def _cpu_busy(interval=0.03, stat='/proc/stat'):
    """{cpu: fraction of `interval` it spent busy}, from two readings of /proc/stat."""
    import time

    def read():
        out = {}
        for line in open(stat):
            if line.startswith('cpu') and line[3].isdigit():
                f = line.split()
                v = [int(x) for x in f[1:]]
                idle = v[3] + (v[4] if len(v) > 4 else 0)
                out[int(f[0][3:])] = (sum(v[:8]) - idle, sum(v[:8]))
        return out
    a = read()
    time.sleep(interval)
    b = read()
    busy = {}
    for c, (bb, bt) in b.items():
        ab, at = a.get(c, (bb, bt))
        busy[c] = (bb - ab) / (bt - at) if bt > at else 0.0
    return busy


def _share_of_node(groups, position, n_peers, busy=None):
    """{'scan', 'io'} for the GPU at `position` among the `n_peers` GPUs attached to a node with these L3 groups:
    the groups are dealt round-robin, the first of a GPU's share runs its scan workers, the others its readers and
    encoders (all of them on the one group when that is all there is).  None when the share is too small to be worth
    pinning to (fewer than four cpus: pinning four scan workers there would serialise them).
    busy ({cpu: busy fraction}, optional): which of the GPU's groups is the quietest decides where the scan workers go."""
    mine = groups[position::n_peers]
    if not mine or len(mine[0]) < 4:
        return None
    if busy and len(mine) > 1:
        # the host is shared: of this GPU's groups the scan workers take the one other tenants use least right now (pinned
        # to cores somebody else keeps busy, a batch ran at half speed)
        load = [sum(busy.get(c, 0.0) for c in g) / len(g) for g in mine]
        best = min(range(len(mine)), key=lambda i: (round(load[i], 2), i))
        mine = [mine[best]] + mine[:best] + mine[best + 1:]
    rest = set().union(*mine[1:]) if len(mine) > 1 else set()
    return {'scan': set(mine[0]), 'io': rest or set(mine[0])}


def _plan_within(allowed, nodes, pci_ids, index, groups_of_node, groups_of_cpus, local_rank=None, busy=None):
    """The plan for GPU `index` inside the cpus this process may use.  First choice: the L3 groups of the GPU's NUMA node,
    shared round-robin with the other GPUs of that node; when none of them is allowed (or the node is unknown), the L3
    groups of whatever is allowed, shared among all GPUs.  Groups with fewer than four allowed cpus do not count.
    local_rank: set when every rank of the node sees only its own GPU (its peers are invisible): the ranks then take the
    groups by their rank on the node."""
    node = nodes[index]
    if node >= 0:
        groups = [g & allowed for g in groups_of_node(node)]
        groups = [g for g in groups if len(g) >= 4]
        if groups and local_rank is not None:
            g = groups[local_rank % len(groups)]
            return {'scan': set(g), 'io': set(g)}
        if groups:
            peers = sorted((i for i in range(len(nodes)) if nodes[i] == node), key=lambda i: pci_ids[i])
            return _share_of_node(groups, peers.index(index), len(peers), busy)
    groups = [g & allowed for g in groups_of_cpus(allowed)]
    groups = [g for g in groups if len(g) >= 4]
    if len(groups) < 2:
        return None                                          # one group (or less) to choose from: nothing to narrow
    return _share_of_node(groups, index % len(groups), max(len(nodes), 1), busy)


def cpu_plan(device=None):
    """Where this process's threads should run, or None to leave them to the scheduler: {'scan': cpus, 'io': cpus}.

    The scan workers hand the interpreter lock and HIP completions back and forth thousands of times per second.  With the
    threads free to roam over a 2 x 64-core host a batch runs at 2.0-2.8 M frames/s (C2); on the eight cores (and their
    SMT siblings) that share one L3 next to the GPU it runs at 2.8-3.6 M; two L3 groups are already worse
    (tools/numa_probe.py).  So: 'scan' = one L3 group of the GPU's NUMA node, 'io' = the other L3 groups of this GPU's share
    of the node (decode readers, encoders: memcpy- and deflate-bound, they should not sit on the scan cores).  GPUs that
    share a NUMA node take its L3 groups round-robin in PCI order, so eight ranks on one host do not overlap.
    Always inside the cpus the process is allowed to use (taskset / numactl / a cpuset only get narrowed, and a mask that
    is one L3 group or less is left alone); SHG_CPU_AFFINITY=off disables it, SHG_CPU_AFFINITY=<cpu list> puts every thread kind on those cpus.

    The first call for a device must come from the thread that owns it (solex_do_work makes it before it starts its
    workers): torch's device queries are not safe to run for the first time from several threads at once.  Whatever goes
    wrong in here means "no placement", never a failed scan."""
    import os
    if not hasattr(os, 'sched_setaffinity'):
        return None
    device = device if device is not None else default_device()
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index in _cpu_plan:
        return _cpu_plan[index]
    plan = None
    choice = os.environ.get('SHG_CPU_AFFINITY', 'auto').strip()
    try:
        if choice.lower() in ('off', '0', 'no', 'none'):
            plan = None
        elif choice.lower() != 'auto':
            cpus = set(_cpulist(choice))
            plan = {'scan': cpus, 'io': cpus} if cpus else None
        else:
            online = set(_cpulist(open('/sys/devices/system/cpu/online').read()))
            allowed = set(os.sched_getaffinity(0)) & online      # a cpuset / taskset is respected: the plan only narrows it
            props = [torch.cuda.get_device_properties(i) for i in range(torch.cuda.device_count())]
            nodes = [int(open('/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node'
                              % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read()) for p in props]
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)
            hidden_peers = local_world > len(props)          # e.g. one visible GPU per rank (HIP_VISIBLE_DEVICES set by a launcher)
            plan = _plan_within(allowed, nodes, [(p.pci_domain_id, p.pci_bus_id, p.pci_device_id) for p in props], index,
                                lambda node: _l3_groups(node, online), lambda cpus: _l3_groups(None, online, cpus=cpus),
                                local_rank=int(os.environ.get('LOCAL_RANK', '0') or 0) if hidden_peers else None, busy=_cpu_busy())
    except Exception:      # noqa: BLE001 -- no sysfs, odd topology, torch without PCI ids: placement is optional
        plan = None
    _cpu_plan[index] = plan
    return plan


def bind_thread(kind, device=None):
    """Put the calling thread (and the threads it creates from now on) on the cpus of `kind` ('scan' or 'io').
    -> the previous mask (to restore with os.sched_setaffinity(0, mask)) or None when nothing was changed."""
    import os
    try:
        plan = cpu_plan(device)
        if not plan:
            return None
        old = os.sched_getaffinity(0)
        os.sched_setaffinity(0, plan[kind])
        return old
    except Exception:      # noqa: BLE001 -- placement is an optimisation, never a reason to fail a scan
        return None


class DeviceImage:
    __array_priority__ = 100

    def __init__(self, tensor, row_factor=None, minmax=None):
        """row_factor (float64 GPU tensor [h], optional): the image is then the float64 array
        tensor[y, x] * row_factor[y] -- what the reference's removeVignette returns (solex_util.py:654) --
        kept factored so that it never has to be materialised in HBM.
        minmax (int32 GPU tensor [2] = {min, max}, optional): the image's extrema as the extraction kernel gathered them
        (shg_extract_columns_minmax); the warp clips to them without another pass over the image."""
        if not isinstance(tensor, torch.Tensor) or not tensor.is_cuda:
            raise TypeError('DeviceImage wraps a GPU tensor')
        self.t = tensor
        self.row_factor = row_factor
        self.minmax = minmax
        self._host = None

    # ndarray surface -------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def dtype(self):
        if self.row_factor is not None:
            return np.dtype(np.float64)
        return np.dtype(str(self.t.dtype).replace('torch.', ''))

    def numpy(self):
        if self._host is None:
            host = self.t.cpu().numpy()
            if self.row_factor is not None:
                host = host * self.row_factor.cpu().numpy().reshape((-1, 1))
            self._host = host
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __len__(self):
        return self.t.shape[0]

    def astype(self, dtype):
        return self.numpy().astype(dtype)

    def __truediv__(self, other):
        return self.numpy() / other

    def __eq__(self, other):
        return self.numpy() == np.asarray(other)

    def __repr__(self):
        return 'DeviceImage(%s, %s, %s)' % (self.shape, self.dtype, self.t.device)


def to_device_u16(img, device=None):
    """GPU uint16 tensor (2-D, unit column stride) from a DeviceImage, tensor or ndarray."""
    if isinstance(img, DeviceImage):
        if img.row_factor is not None:
            raise TypeError('this stage needs a uint16 image, got a factored float64 one')
        return img.t
    if isinstance(img, torch.Tensor):
        if not img.is_cuda:
            raise RuntimeError('expected a GPU tensor')
        return img
    arr = np.asarray(img)
    if arr.dtype != np.uint16:
        raise TypeError('expected a uint16 image, got %s' % arr.dtype)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device or default_device())


def u16_from_unit_float(image):
    """The reference hands `disk / 65536` (float64) to correct_image (Solex_recon.py:123,
    ellipse_to_circle.py:299).  Recover the exact uint16 disk, or refuse."""
    if isinstance(image, (DeviceImage, torch.Tensor)):
        return image
    arr = np.asarray(image)
    if arr.dtype == np.uint16:
        return arr
    scaled = arr * 65536.0
    back = np.rint(scaled)
    if arr.size and (np.abs(scaled - back).max() != 0 or back.min() < 0 or back.max() > 65535):
        raise ValueError('correct_image expects uint16 data or uint16/65536 (16-bit assumption of the reference)')
    return back.astype(np.uint16)
