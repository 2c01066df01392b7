"""K1 / K2 at the C2 shape, 20 launches each (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys

sys.path.insert(0, "gnn-tumor-seg_amd")
import torch  # noqa: E402

import gts  # noqa: E402
from gts import ops, synth  # noqa: E402

g = gts.batch([synth.lattice_graph() for _ in range(4)]).to("cuda")
x = torch.randn(g.n, 256, device="cuda").relu_()
gout = torch.randn(g.n, 256, device="cuda")
big = torch.empty(128 << 20, device="cuda")          # 512 MB: written between launches so that nothing stays cached
for _ in range(20):
    out, arg = ops.spmm_max_fwd(g, x, relu_input=True)
    big.fill_(1.0)
    gx = ops.spmm_max_bwd(g, gout, arg)
    big.fill_(2.0)
torch.cuda.synchronize()
print("done")
