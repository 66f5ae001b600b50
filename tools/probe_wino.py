"""Per-launch time of the 33 TFC 3x3 convs of one 4-minute demix (HQ_3 layout, synthetic weights), direct kernel against the
Winograd kernels (WINO = 0 / 1 / 2 / 3) and their measurement-only ablation builds (ASX_WINO_ABL, results invalid)."""
import os
import sys

import torch

sys.path.insert(0, ".")
import audio_separator_amd as A
from oracle import mdx_oracle as O

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
d = O.NetDims()
eng = A.Engine(A.MDXConfig(max_batch=64))
eng.set_option("winograd", int(os.environ.get("WINO", "0")))
eng.load_net(A.NetConfig(), A.fold_convtdf_state(O.make_convtdf_state(d, seed=0), d.num_blocks, d.l))
n = int(44100 * secs)
mix = torch.from_numpy(O.synth_mix(n, seed=0)).cuda()
out = torch.empty_like(mix)
st = torch.cuda.current_stream().cuda_stream
eng.demix_dev(mix.data_ptr(), n, out.data_ptr(), stream=st)
torch.cuda.synchronize()
eng.profile_enable(True)
eng.demix_dev(mix.data_ptr(), n, out.data_ptr(), stream=st)
allrecs = eng.profile_launches()
recs = [r for r in allrecs if r[0] == "conv3x3"]
eng.profile_enable(False)
if os.environ.get("TDF"):
    t = [r for r in allrecs if r[0] == "tdf"]
    print("tdf total %.2f ms;" % sum(r[1] for r in t), "per launch (ms | TF/s):", " ".join(f"{r[1]:.2f}|{r[2] / r[1] / 1e9:.0f}" for r in t), flush=True)
tot = sum(r[1] for r in recs)
tag = f"WINO={os.environ.get('WINO', '0')} ABL={os.environ.get('ASX_WINO_ABL', '0')}"
print(tag, f"conv3x3 total {tot:.2f} ms;", "per launch (ms | direct-equivalent TF/s):",
      " ".join(f"{r[1]:.2f}|{r[2] / r[1] / 1e9:.0f}" for r in recs), flush=True)
