"""Probe: duration of the small-M conv kernel as a function of K (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd.ops import Ctx, Var
from nvae_tf_amd import ops
from nvae_tf_amd.params import ParamStore
dev = torch.device("cuda:0")
for cin in (64, 128, 256, 512):
    ps = ParamStore(seed=1)
    conv = ps.conv("c", 3, cin, 256)
    ps.finalize(dev, torch.bfloat16, zero_pool_floats=1 << 16)
    ps.begin_step(); ps.prepare_weights(False)
    x = Var(torch.randn(128, 4, 4, cin, device=dev).bfloat16())
    for _ in range(20):
        ctx = Ctx(ps, torch.bfloat16, True, False)
        ops.conv2d(ctx, x, conv)
    torch.cuda.synchronize()
