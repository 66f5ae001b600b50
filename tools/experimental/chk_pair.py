import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import mdx_oracle as O
import audio_separator_amd as A
d = O.NetDims()
sd = O.make_convtdf_state(d, seed=3)
eng = A.Engine(A.MDXConfig(max_batch=2))
eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
mix = O.synth_mix(600_000, seed=1)
p0, h0 = eng.counter("tdf3_pair_image_launches"), eng.counter("tdf3h_launches")
eng.demix(mix)
print("pair", eng.counter("tdf3_pair_image_launches") - p0, "tdf3h", eng.counter("tdf3h_launches") - h0, "plan", eng.plan(600_000))
