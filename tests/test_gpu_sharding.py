"""The chunk-range / fold halves of the sibling loops on the GPU: running every chunk through
sharding.sharded_demix (single process: ranges -> chunks -> finalize) must reproduce the monolithic *_demix_dev call
bit for bit, for any split of the chunk list."""
from fractions import Fraction

import numpy as np
import pytest

from oracle import demucs_oracle as D
from oracle import mdxc_oracle as M
from oracle import roformer_oracle as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import audio_separator_amd as A
    return A


def _split_run(adapter, mix, torch, parts):
    """chunks computed in `parts` separate range calls, then one fold"""
    n = mix.shape[-1]
    if hasattr(adapter, "bind_mix"):
        adapter.bind_mix(mix)
    plan = adapter.plan(n)
    nk, C = plan["n_chunks"], plan["chunk_size"]
    allc = torch.zeros((nk, adapter.stems, 2, C), dtype=torch.float32, device="cuda")
    from audio_separator_amd.sharding import partition_chunks
    for k0, k1 in partition_chunks(nk, parts):
        if k1 > k0:
            adapter.demix_chunks(mix, n, k0, k1, allc[k0:k1])
    out = torch.empty((adapter.out_stems, 2, n), dtype=torch.float32, device="cuda")
    adapter.finalize(allc, n, out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_roformer_halves(A):
    import torch
    from audio_separator_amd.sharding import RoformerAdapter, sharded_demix
    cfg = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64, stft_hop_length=16,
                           stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 2},
                       state_dict=R.make_roformer_state(cfg, 7), max_batch=3)
    mixh = (0.4 * np.random.default_rng(3).standard_normal((2, 1700))).astype(np.float32)
    want = dm.engine.rof_demix(mixh, 200)
    mix = torch.from_numpy(mixh).cuda()
    ad = RoformerAdapter(dm.engine, 200)
    assert np.array_equal(sharded_demix(ad, mix).cpu().numpy(), want)
    assert np.array_equal(_split_run(ad, mix, torch, 3), want)


def test_mdxc_halves(A):
    import torch
    from audio_separator_amd.sharding import MdxcAdapter, sharded_demix
    cfg = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=1,
                     num_channels_model=8, growth=8, bottleneck_factor=2)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": 4}, state_dict=M.make_v3_state(cfg, 2),
                       max_batch=3)
    mixh = (0.4 * np.random.default_rng(4).standard_normal((2, 1500))).astype(np.float32)
    want = dm.engine.mdxc_demix(mixh, 4)
    mix = torch.from_numpy(mixh).cuda()
    ad = MdxcAdapter(dm.engine, 4)
    assert np.array_equal(sharded_demix(ad, mix).cpu().numpy(), want)
    assert np.array_equal(_split_run(ad, mix, torch, 4), want)


def test_demucs_halves(A):
    import torch
    from audio_separator_amd.sharding import DemucsAdapter, sharded_demix
    oc = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2, samplerate=8000, segment=Fraction(1, 1))
    hc = A.HTConfig(sources=tuple(oc.sources), channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2,
                    samplerate=8000, segment=Fraction(1, 1), max_batch=3)
    eng = A.Engine(A.MDXConfig(n_fft=1024, hop_length=256, dim_f=512, segment_size=8))
    eng.load_ht(hc, D.make_ht_state(oc, 11))
    mixh = (0.3 * np.random.default_rng(5).standard_normal((2, 20011)) + 0.01).astype(np.float32)
    offs = [1234, 77]
    want = eng.ht_demix(mixh, shifts=2, offsets=offs, overlap=0.25, standardize=True, swap01=True)
    mix = torch.from_numpy(mixh).cuda()
    ad = DemucsAdapter(eng, shifts=2, offsets=offs, overlap=0.25, flags=3)
    assert np.array_equal(sharded_demix(ad, mix).cpu().numpy(), want)
    assert np.array_equal(_split_run(ad, mix, torch, 5), want)


def test_hdemucs_halves(A):
    # Demucs v3: the chunk forwards of a call (each at its own length) split across ranks exactly like the v4 segments
    import torch
    from oracle import hdemucs_oracle as H
    from audio_separator_amd.sharding import DemucsAdapter, sharded_demix
    oc = H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)
    hc = A.HDConfig(sources=tuple(oc.sources), channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000,
                    segment=2, max_batch=2)
    eng = A.Engine(A.MDXConfig(n_fft=1024, hop_length=256, dim_f=512, segment_size=8))
    eng.load_hd(hc, H.make_hd_state(oc, 21))
    mixh = (0.3 * np.random.default_rng(6).standard_normal((2, 50011)) + 0.01).astype(np.float32)
    offs = [1234, 77]
    want = eng.hd_demix(mixh, shifts=2, offsets=offs, overlap=0.25, standardize=True, swap01=True)
    mix = torch.from_numpy(mixh).cuda()
    ad = DemucsAdapter(eng, shifts=2, offsets=offs, overlap=0.25, flags=3, v3=True)

    def close(a, b):   # float64 atomics in the statistics: the last bit may depend on the batch composition
        return np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) < 1e-6 * np.sqrt(np.mean(b.astype(np.float64) ** 2))
    assert close(sharded_demix(ad, mix).cpu().numpy(), want)
    assert close(_split_run(ad, mix, torch, 3), want)


def test_shard_workspace_graph_replay_is_bit_identical(A):
    """ShardWorkspace(graph=True): the rank's chunk-range compute captured into a hipGraph on its second call and replayed from then on (VERDICT r5
    #7b: 130 launches in a 28-ms strong-scaling step is where host launch jitter shows first).  The replayed passes equal the eager one bit for bit,
    on the HQ_3 geometry's fast FFT path with a real (small) net, for new input contents in the same buffers."""
    import torch
    from oracle import mdx_oracle as O
    from audio_separator_amd.sharding import HipEngineAdapter, ShardWorkspace, sharded_demix
    d = O.NetDims()
    eng = A.Engine(A.MDXConfig(max_batch=4))
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(O.make_convtdf_state(d, seed=0), d.num_blocks, d.l))
    n = 44100 * 20
    mix = torch.from_numpy(O.synth_mix(n, seed=1)).cuda()
    ad = HipEngineAdapter(eng)
    eager = sharded_demix(ad, mix, workspace=ShardWorkspace()).clone()
    ws = ShardWorkspace(graph=True)
    outs = [sharded_demix(ad, mix, workspace=ws).clone() for _ in range(4)]
    torch.cuda.synchronize()
    assert ws.graph_error is None and ws.graph_replays == 3, (ws.graph_error, ws.graph_replays)
    for o in outs:
        assert torch.equal(o, eager)
    mix.mul_(0.5)                                       # same buffers, new content
    again = sharded_demix(ad, mix, workspace=ws).clone()
    torch.cuda.synchronize()
    assert ws.graph_replays == 4
    assert torch.equal(again, sharded_demix(ad, mix, workspace=ShardWorkspace()))
    eng.close()
