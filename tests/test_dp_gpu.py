"""Data-parallel path with REAL HIP kernels and two ranks (`-m gpu`, one MI355X): two processes share cuda:0, each runs
the product trainer (HipOps, teacher stream + weight-gradient stream + communication stream, small buckets) on its
shard of the batch and exchanges gradients through `GradReducer` over the gloo backend on device tensors (RCCL refuses
two ranks on one device; the 8-GPU run uses the same code with backend "nccl").  What this covers that the CPU gloo
test and the RCCL world-size-1 test cannot: the stream hand-off main -> weight-gradient stream -> communication stream
-> optimizer with asynchronous kernels and a second rank.  Reference: accelerator.prepare -> DDP, run_distillation.py:
1449-1451, backward with bucketed all-reduce 1609."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import whisper_oracle as wo

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=17):
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(4, cfg_t.n_mels, 3000, generator=g) * 0.5
    b = wo.synthetic_batch(cfg_t, 4, seed=seed + 1, T=48, with_audio=False)
    return cfg_t, cfg_s, t_sd, s_sd, feats, b["decoder_input_ids"], b["labels"]


def _trainer(cfg_t, cfg_s, t_sd, s_sd, **kw):
    from distil_whisper_amd.distill import DistillationTrainer
    from distil_whisper_amd.ops_hip import HipOps
    filt = torch.tensor(wo.mel_filter_bank(cfg_t.n_mels), dtype=torch.float32).cuda().contiguous()
    return DistillationTrainer(HipOps("cuda:0"), s_sd, cfg_s, t_sd, cfg_t, mel_filters=filt, weight_decay=0.05, **kw)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_t, cfg_s, t_sd, s_sd, feats, ids, labels = _make()
    tr = _trainer(cfg_t, cfg_s, t_sd, s_sd, overlap_teacher=True, overlap_wgrad=True, bucket_bytes=64 << 10)
    assert tr.world == 2 and tr.reducer.active and tr.reducer.stream is not None
    assert tr.reducer.also_wait == [tr.student.wgrad_stream]
    launched = []
    orig = tr.reducer._launch
    tr.reducer._launch = lambda lo, hi: launched.append((lo, hi)) or orig(lo, hi)
    sl = slice(rank * 2, rank * 2 + 2)
    f, i, l = feats[sl].cuda(), ids[sl].cuda(), labels[sl].cuda()
    losses = []
    for _ in range(3):
        losses.append(tr.train_step(f, i, l).clone())
    torch.cuda.synchronize()
    st = tr.student_store
    torch.save({"P": st.P.cpu(), "S": st.S.cpu(), "world": tr.world, "buckets": len(launched),
                "lo": min(a for a, _ in launched), "hi": max(b for _, b in launched),
                "range": (st.train_start, st.train_end), "losses": torch.stack(losses).cpu(),
                "step": tr.step_count}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_stay_identical_and_match_the_averaged_gradient_run(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert r0["world"] == r1["world"] == 2 and r0["step"] == r1["step"] == 3
    assert r0["buckets"] >= 12 and (r0["lo"], r0["hi"]) == r0["range"]       # >= 4 buckets per step, whole range covered
    assert torch.equal(r0["P"], r1["P"]) and torch.equal(r0["S"], r1["S"])     # replicas stay bit-identical
    # one process, the two shards as micro-batches: the fused AdamW averages them exactly like the ranks' all-reduce
    cfg_t, cfg_s, t_sd, s_sd, feats, ids, labels = _make()
    tr = _trainer(cfg_t, cfg_s, t_sd, s_sd)
    feats, ids, labels = feats.cuda(), ids.cuda(), labels.cuda()
    for step in range(3):
        out = tr.train_step_accumulated([(feats[0:2], ids[0:2], labels[0:2]), (feats[2:4], ids[2:4], labels[2:4])])
        both = 0.5 * (r0["losses"][step] + r1["losses"][step])
        assert abs(out[2].item() - both[2].item()) < 1e-4 * abs(both[2].item()), step
    torch.cuda.synchronize()
    a, b = tr.student_store.P.cpu(), r0["P"]
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 5e-6, rel


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it (how the driver calls it) starts two ranks under
    torch.distributed.run, runs the data-parallel step and the replica check, and rank 0 prints ONE JSON line for the
    whole job.  Both ranks share cuda:0 over gloo here (one GPU on the test box); the 8-GPU run differs in the backend
    argument only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--model", "tiny.en", "--backend", "gloo", "--share-device", "--no-roofline",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["step_mode"].startswith("eager")
    assert "replica check: parameters bit-identical on all ranks" in r.stderr
    # and a request for more GPUs than the box has fails loudly instead of silently running one rank
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "1"],
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "GPU(s) are visible" in r2.stderr
