"""How long does a cold microbenchmark of k_conv_halo stay representative?  Launches the dominant 5x5 layer
back to back from an idle chip and prints the mean launch time of consecutive windows of 20 launches (HIP events on the
launch stream) next to the shader clock / socket power rocm-smi reports at that moment.
usage: python tools/mb_dvfs.py [windows]"""
import ctypes as C, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L
from nvae_tf_amd.ops import same_pad

dev = "cuda:0"
lib = L.load()
B, H, ci, co, k = 128, 16, 384, 384, 5
p = same_pad(H, k, 1)[0]
g = L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, p, p, 1, 0, ci, co, co)
x = torch.randn(B, H, H, ci, device=dev).bfloat16()
w = (torch.randn(co, k * k * ci, device=dev) * 0.05).bfloat16()
out = torch.empty(B, H, H, co, device=dev, dtype=torch.bfloat16)
fn = lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None)
fn(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        fn()


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(o)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower()}
    except Exception as e:      # noqa
        return {"rocm-smi": repr(e)[:80]}


n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 400
print("idle:", smi(), flush=True)
time.sleep(2.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
ev[0].record()
for i in range(n_win):
    gr.replay()
    ev[i + 1].record()
t_smi = smi()                      # sampled while the queue is still draining
torch.cuda.synchronize()
us = [ev[i].elapsed_time(ev[i + 1]) * 1000 / 20 for i in range(n_win)]
flop = 2.0 * B * H * H * co * k * k * ci
cum = 0.0
for i in (0, 1, 2, 3, 5, 8, 12, 20, 30, 50, 80, 120, 200, 300, n_win - 1):
    if i < n_win:
        print(f"window {i:4d} (t = {sum(us[:i]) * 20 / 1000:7.1f} ms): {us[i]:6.1f} us/launch = {flop / us[i] / 1e6:7.1f} TFLOP/s")
print("under load:", t_smi)

# --- the training step's duty pattern: short MFMA bursts between phases of small kernels.  Graph A = 6 launches of the
# 5x5 layer + 300 tiny elementwise kernels (~1.5 ms), graph B = the 300 tiny kernels alone; both replayed 100 times back
# to back; (A - B) / 6 is what a launch of the big kernel costs inside that pattern.
small = torch.randn(1 << 16, device=dev).bfloat16()
small_o = torch.empty_like(small)
tiny = lambda: L.call("nvae_unary_fwd", L.BF16, 2, L.ptr(small), L.ptr(small_o), small.numel(), 1.0, 0.0)
def run(gr, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


tiny(); torch.cuda.synchronize()
for n_tiny in (300, 1500, 6000):
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        for _ in range(6):
            fn()
        for _ in range(n_tiny):
            tiny()
    with torch.cuda.graph(gb):
        for _ in range(n_tiny):
            tiny()
    time.sleep(0.5)
    reps = max(30000 // n_tiny, 10)
    for rnd in range(2):
        tb = run(gb, reps); ta = run(ga, reps)
        per = (ta - tb) / 6
        print(f"duty pattern, 6 big + {n_tiny:5d} tiny kernels, round {rnd}: A {ta:9.1f} us, B {tb:9.1f} us ({tb / n_tiny:5.2f} us per tiny kernel) -> "
              f"{per:6.1f} us per 5x5 launch = {flop / per / 1e6:7.1f} TFLOP/s (MFMA duty {6 * per / ta * 100:4.1f} %)", flush=True)
