import os, sys
sys.path.insert(0, os.getcwd())
from ug_stereomatcher_amd import _lib
c = _lib.Context(levels=5)
p = c.alloc(1024)
c.free(p)
import torch
try:
    t = torch.zeros(4, device="cuda")
    print("torch after libugsm: ok", t.sum().item())
except Exception as e:
    print("torch after libugsm: FAILED:", e)
import subprocess
print(subprocess.run("cat /proc/%d/maps | grep -i 'amdhip\|hsa-runtime' | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
c.close()
