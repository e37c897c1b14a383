"""Does keeping the process on a few cores next to the GPU steady the pipelined bench?  Prints the topology and runs
bench.py --steps 50 five times per setting of SHG_CPU_AFFINITY (device.cpu_plan)."""
import glob
import json
import os
import subprocess
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for n in sorted(glob.glob('/sys/devices/system/node/node*')):
    print(os.path.basename(n), open(n + '/cpulist').read().strip())
props = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id) if hasattr(props, 'pci_bus_id') else None
print('gpu', props.name, 'bdf', bdf)
node = None
if bdf and os.path.exists('/sys/bus/pci/devices/%s/numa_node' % bdf):
    node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
print('gpu numa node', node)


def run(env_extra, tag):
    vals = []
    for _ in range(5):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-e2e', '--steps', '50'], env=env,
                             capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith('{')][-1]
        vals.append(json.loads(line)['value'])
    print(tag, ['%.2fM' % (v / 1e6) for v in vals])


def cpus_of(n):
    out = []
    for part in open('/sys/devices/system/node/node%d/cpulist' % n).read().strip().split(','):
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


near = cpus_of(node if node is not None and node >= 0 else 0)
far = cpus_of(1 - node) if node in (0, 1) and os.path.exists('/sys/devices/system/node/node1') else near
fmt = lambda c: ','.join(str(v) for v in c)      # noqa: E731
run({'SHG_CPU_AFFINITY': 'off'}, 'unbound                ')
run({}, 'auto                   ')
for label, cpus in (('near node, 4 cores ', near[:4]), ('near node, 8 cores ', near[:8]), ('near node, 8 cores #2', near[8:16]), ('near node, 16 cores', near[:16]),
                    ('far node, 8 cores  ', far[:8]), ('near node, 8 cores + SMT', near[:8] + near[len(near) // 2:len(near) // 2 + 8])):
    run({'SHG_CPU_AFFINITY': fmt(cpus)}, '%-24s %s' % (label, fmt(cpus)[:40]))
