#!/bin/bash
out=gpurun_out/r03t
mkdir -p $out
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|MHz" 
for d in /sys/class/drm/card*/device; do echo "$d numa=$(cat $d/numa_node 2>/dev/null) cpus=$(cat $d/local_cpulist 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done 2>/dev/null | head -20
python3 - <<'PY'
import torch
print(torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else "")
PY
rocm-smi --showbus 2>/dev/null | head -12
rocm-smi --showtoponuma 2>/dev/null | head -12
cat /proc/self/status | grep -i "mems_allowed_list\|cpus_allowed_list"
