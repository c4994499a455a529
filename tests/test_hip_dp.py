"""Data-parallel path on the GPU box: the overlapped gradient exchange must start only when the decoder's gradients
are final (the reference calls its decoder with keyword arguments, model/model.py:298-306, which is exactly the case a
module-level backward hook gets wrong), the two-rank rehearsal must reproduce the mean of the per-rank gradients, and
FlatAdam must accept Lightning's closure."""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(cfg_name, seed, dropout=True):
    from oracle import model_config, fill_state
    from transformertts_amd.model import TransformerTTS
    from transformertts_amd.model.layers import MultiheadAttention
    cfg = model_config(cfg_name)
    m = TransformerTTS(**cfg, device="cuda")
    m.load_state_dict(fill_state(cfg, seed), strict=True)
    m = m.to("cuda")
    if not dropout:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, MultiheadAttention):
                mod.dropout = 0.0
    return cfg, m


@pytest.mark.parametrize("cfg_name,B,Tp,Tm", [("tiny", 3, 12, 40), ("base", 2, 50, 200)])
def test_tail_trigger_fires_after_decoder_gradients_are_final(cfg_name, B, Tp, Tm):
    """Snapshot bucket.flat[lo:] at the moment the overlap trigger fires (stream-ordered clone) and require it to be
    bit-equal to the same range after backward() has returned: nothing in the tail may still be written afterwards.
    Any PyTorch warning (e.g. 'Full backward hook is firing when gradients are computed with respect to module
    outputs') is an error."""
    from oracle import synth_batch
    from transformertts_amd.loss import TransformerTTSLoss
    from transformertts_amd.parallel import FlatGradBucket, overlap_tail_with_backward
    cfg, m = _model(cfg_name, 5)
    m.train()
    bucket = FlatGradBucket(m.parameters())
    snaps = []
    trig = overlap_tail_with_backward(bucket, m, m.decoder, on_ready=lambda lo: snaps.append((lo, bucket.flat[lo:].clone())))
    assert trig is not None
    lo = trig.lo
    assert lo == bucket.offset_of(next(m.decoder.parameters())) and 0 < lo < bucket.flat.numel()
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=9)
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    crit = TransformerTTSLoss(8.0).to("cuda")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for it in range(2):                       # twice: the trigger re-arms on every grad-enabled forward
            bucket.zero()
            with torch.no_grad():                 # forward #1 of training_step: must not arm or fire anything
                m(*args, need_alignments=False)
            loss = crit(m(*args), args[1], args[3])["total"]
            loss.backward()
            torch.cuda.synchronize()
            assert len(snaps) == it + 1 and trig.fired == it + 1
            at_fire_lo, at_fire = snaps[-1]
            assert at_fire_lo == lo
            assert torch.equal(at_fire, bucket.flat[lo:]), "tail gradients changed after the overlap trigger fired"
            assert float(at_fire.abs().sum()) > 0
            # and the head (encoder side) was NOT yet complete at that point in a meaningful sense: it is non-zero now
            assert float(bucket.flat[:lo].abs().sum()) > 0
    trig.remove()


def test_flat_adam_accepts_lightning_style_closure():
    """optimizer.step(closure) with a closure that runs zero_grad + forward + backward (Lightning's automatic
    optimisation): same trajectory as torch.optim.Adam driven by the same closure."""
    from oracle import synth_batch
    from transformertts_amd.loss import TransformerTTSLoss
    from transformertts_amd.optim import FlatAdam
    cfg, m1 = _model("tiny", 7, dropout=False)
    _, m2 = _model("tiny", 7, dropout=False)
    batch = synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=3)
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    crit = TransformerTTSLoss(8.0).to("cuda")
    o1 = FlatAdam(m1.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)

    def closure_for(m, o):
        def closure():
            assert torch.is_grad_enabled()
            o.zero_grad()
            m.train()
            loss = crit(m(*args), args[1], args[3])["total"]
            loss.backward()
            return loss
        return closure
    l1 = l2 = None
    for _ in range(3):
        l1 = o1.step(closure_for(m1, o1))
        l2 = o2.step(closure_for(m2, o2))
    assert l1 is not None and abs(l1.item() - l2.item()) < 1e-4 * abs(l2.item())
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        # analytically zero gradients (a bias in front of BatchNorm, the key-bias third of a packed in-projection bias:
        # softmax is shift-invariant): Adam (eps 1e-9) turns their rounding noise into +-lr steps in both optimizers
        if p2.grad.norm().item() < 1e-6 or n.endswith("in_proj_bias"):
            continue
        assert rel_l2(p1, p2) < 1e-4, (n, rel_l2(p1, p2))


def test_weight_planes_follow_raw_data_writes_after_epoch_bump():
    """The cached bf16 hi/mid/lo planes of a weight are keyed on autograd's version counter; a raw `.data` write (what
    broadcast_module_state, EMA or manual weight loading do) does not move it, so the writer must call
    ops.bump_param_epoch() -- after which the split GEMM sees the new values."""
    from transformertts_amd import ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 256).to("cuda")
    x = torch.randn(300, 256, device="cuda")
    with torch.no_grad():
        y0 = ops.linear(x, lin.weight, lin.bias)
        lin.weight.data.mul_(2.0)
        ops.bump_param_epoch()
        y1 = ops.linear(x, lin.weight, lin.bias)
    ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    assert rel_l2(y1, ref) < 1e-6, rel_l2(y1, ref)
    assert rel_l2(y0, ref) > 1e-2          # the first result was computed with the old weights


@pytest.mark.parametrize("overlap,graph", [("1", "0"), ("0", "0"), ("1", "1")])
def test_two_rank_rehearsal_reduces_to_the_mean_gradient(overlap, graph):
    """bench.py in rehearsal mode (two gloo ranks sharing cuda:0): the reduced bucket must equal the mean of the two
    ranks' single-process gradients (rel-L2 <= 1e-6), with the overlapped tail exchange on and off.  The command is the
    bare `python bench.py --gpus 2 ...`: bench.py starts its own torch.distributed.run child (bench.self_launch)."""
    env = dict(os.environ, TTTS_BENCH_REHEARSAL="1", TTTS_DP_OVERLAP=overlap, TTTS_GRAPH=graph,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "3" if graph == "1" else "1", "--batch", "4", "--tm", "160", "--tp", "40", "--ragged", "--no-cpu-baseline",
           "--no-probe"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2
    chk = out["config"]["rehearsal_gradient_check"]
    assert chk["overlap_requested"] == (overlap == "1")
    assert chk["tail_trigger_fired"] == (overlap == "1")
    assert chk["rel_l2_vs_mean_of_rank_gradients"] <= 1e-6, chk
    assert "Full backward hook is firing" not in r.stderr


def test_rccl_one_rank_group_matches_the_single_process_step():
    """RCCL itself (backend "nccl"), on the one GPU of this box: a one-rank process group launched by torch.distributed.run
    as the driver launches bench.py.  tools/rccl_one_rank.py takes the data-parallel launch path -- graph replay, then
    ReduceOp.AVG over the flat bucket on the same stream, then clip + Adam -- and requires the one-rank mean to leave the
    bucket unchanged and six steps (dropout on) to end bit for bit where the single-process path ends."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "tools", "rccl_one_rank.py"), "base", "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "rccl_one_rank.json"), "w") as f:
        json.dump(out, f)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["one_rank_mean_leaves_bucket_unchanged"] and out["losses_equal"] and out["state_equal"], out
    # ... and with the exchange overlapped on the FAST path: the step captured as two graphs cut where backward leaves the decoder,
    # the tail's all-reduce (RCCL, asynchronous) started between their replays -- same bits again
    assert out["split_graphs"] == 1 and out["overlap_losses_equal"] and out["overlap_state_equal"], out
    # none of the three un-sabotaged runs may have taken the guarded fallback (one graph instead of two, or eager launches)
    assert out["normal_path_fallbacks"] == [None, None, None], out
    # ... and when the two-graph capture cannot be made (the tail graph's capture_begin is made to raise): one graph per step, said
    # so on stderr and in `capture_fallback`, same bits
    assert out["fallback"] and out["fallback_split_graphs"] == 0 and out["fallback_losses_equal"] and out["fallback_state_equal"], out
    assert "two-graph capture failed" in r.stderr, r.stderr[-2000:]


def test_bench_runs_under_a_one_rank_rccl_group():
    """bench.py itself on the data-parallel launch path over the real backend: RANK / WORLD_SIZE = 1 with TTTS_FORCE_DIST=1 makes
    it initialise "nccl" (RCCL), broadcast the module state, capture the step as two graphs cut at the tail trigger and
    exchange the bucket every step -- what every rank of an N-GPU run does, minus the other ranks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", TTTS_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "6", "--warmup", "3", "--sustain", "0", "--batch", "16",
                        "--no-cpu-baseline", "--no-alignments-figure"], env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["process_group"].startswith("nccl")
    assert out["config"]["grad_allreduce"] == "tail overlapped with backward", out["config"]
    assert "failed" not in out["config"]["launch_path"], out["config"]["launch_path"]
