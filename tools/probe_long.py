"""Long-input sanity of the four loops (memory and index arithmetic at sizes the tests do not reach): a 20-minute song through the
MDX engine, 10 minutes through htdemucs / VR, 3 minutes through BS-Roformer.  Prints RTF, finiteness and the peak device memory."""
import sys
import time
from fractions import Fraction

import numpy as np
import torch

sys.path.insert(0, ".")
import audio_separator_amd as A
from oracle import mdx_oracle as O

SR = 44100


def run(name, secs, step, out):
    step()
    torch.cuda.synchronize()
    t0 = time.time()
    step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    free, total = torch.cuda.mem_get_info()
    print(f"{name}: {secs:.0f} s in {dt*1e3:.1f} ms = {secs/dt:.1f}x RT, finite {bool(torch.isfinite(out).all())}, device memory in use {(total-free)/2**30:.1f} GiB", flush=True)


st = torch.cuda.current_stream().cuda_stream
d = O.NetDims()
eng = A.Engine(A.MDXConfig())
eng.load_net(A.NetConfig(), A.fold_convtdf_state(O.make_convtdf_state(d, seed=0), d.num_blocks, d.l))
n = SR * 1200
mix = torch.from_numpy(O.synth_mix(n, seed=0)).cuda()
out = torch.empty_like(mix)
run("mdx_hq3", 1200, lambda: eng.demix_dev(mix.data_ptr(), n, out.data_ptr(), stream=st), out)
eng.close()
del mix, out
torch.cuda.empty_cache()

from oracle import demucs_oracle as D
oc = D.HTConfig()
eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
eng.load_ht(A.HTConfig(segment=Fraction(39, 5)), D.make_ht_state(oc, 0))
n = SR * 600
mix = torch.from_numpy(O.synth_mix(n, seed=1)).cuda()
out = torch.empty((4, 2, n), dtype=torch.float32, device="cuda")
run("htdemucs", 600, lambda: eng.ht_demix_dev(mix.data_ptr(), n, out.data_ptr(), shifts=2, offsets=[11025, 3000], flags=3, stream=st), out)
eng.close()
del mix, out
torch.cuda.empty_cache()

from oracle import vr_oracle as V
sys.path.insert(0, "tools")
from fullsong_cases import VR_MP
arch = 123821
dm = A.VRDemixer({"model_params": VR_MP, "primary_stem_name": "Instrumental", "torch_device": 0},
                 {"window_size": 512, "batch_size": 4, "aggression": 5}, state_dict=V.make_vr_state(arch, 0), nn_arch_size=arch)
n = SR * 600
w = torch.from_numpy(O.synth_mix(n, seed=2)).cuda()
_, n_out = dm.engine.vr_plan(n)
p = torch.empty((2, n_out), dtype=torch.float32, device="cuda")
s = torch.empty_like(p)
run("vr", 600, lambda: dm.engine.vr_separate_dev(w.data_ptr(), n, p.data_ptr(), s.data_ptr(), 0.05, 186, stream=st), p)
