"""Development probe (GPU box only): where does the GPU sit, which CPUs are local to it, where does this process run."""
import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
p = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
print('gpu', p.name, bdf)
for f in ('local_cpulist', 'numa_node'):
    try:
        print(f, open('/sys/bus/pci/devices/%s/%s' % (bdf, f)).read().strip())
    except Exception as ex:
        print(f, 'unavailable', ex)
print('affinity', sorted(os.sched_getaffinity(0))[:8], '... n =', len(os.sched_getaffinity(0)))
print('nproc', os.cpu_count())
for node in sorted(glob.glob('/sys/devices/system/node/node*/cpulist')):
    print(node, open(node).read().strip())
