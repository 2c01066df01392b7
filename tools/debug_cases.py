"""Debug helper: run single kernel cases in child processes so a GPU fault is isolated."""
import subprocess
import sys

CASE = r'''
import sys, numpy as np, torch
sys.path.insert(0, "gnn-tumor-seg_amd"); sys.path.insert(0, ".")
import gts
from gts import ops
from tests.helpers import random_coo, ref_and_gts, slots_to_sources
from oracle import torch_ref
f, want_arg = int(sys.argv[1]), int(sys.argv[2])
n = 300
src, dst = random_coo(n, 2000, seed=f)
tg, g = ref_and_gts(src, dst, n)
x = torch.randn(n, f)
out, arg = ops.spmm_max_fwd(g.to("cuda"), x.cuda(), want_arg=bool(want_arg))
torch.cuda.synchronize()
ref, aref = torch_ref.spmm_max_with_arg(tg, x)
print("f", f, "arg", want_arg, "equal", torch.equal(out.cpu(), ref))
'''

for f in (260, 300, 512, 516):
    for want_arg in (0, 1):
        r = subprocess.run([sys.executable, "-c", CASE, str(f), str(want_arg)], capture_output=True, text=True, timeout=120)
        print(f"--- f={f} arg={want_arg} rc={r.returncode}")
        print(r.stdout[-300:])
        if r.returncode != 0:
            print(r.stderr[-600:])
