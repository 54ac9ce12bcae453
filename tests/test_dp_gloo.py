"""Data-parallel path on CPU: world_size 2 over gloo, one process per rank (the GPU run uses the same code over RCCL).
Checks DDP semantics of the reference (per-rank token-mean loss, gradients averaged over ranks: SURVEY.md 2.2 C1):
parameters stay bit-identical across ranks and match a single process that averages the two ranks' gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distil_whisper_amd.distill import DistillationTrainer
from oracle import whisper_oracle as wo
from oracle.ref_ops import RefOps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=7):
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(4, cfg_t.n_mels, 3000, generator=g) * 0.5
    b = wo.synthetic_batch(cfg_t, 4, seed=seed + 1, T=33, with_audio=False)
    labels = b["labels"].clone()
    for i, n in enumerate(LENS):                 # dead tails of different lengths (rank 0: 20, 9; rank 1: 27, 14)
        labels[i, n:] = -100
    return cfg_t, cfg_s, t_sd, s_sd, feats, b["decoder_input_ids"], labels


LENS = [20, 9, 27, 14]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_t, cfg_s, t_sd, s_sd, feats, ids, labels = _make()
    tr = DistillationTrainer(RefOps("cpu", lowp=torch.float32), s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.05)
    tr.reducer.bucket_elems = 20000  # several buckets
    sl = slice(rank * 2, rank * 2 + 2)
    for _ in range(2):
        # every rank leaves out ITS dead decoder positions (different trimmed lengths and packed row counts per rank);
        # the single-process reference below computes all 33 positions
        tr.train_step(feats[sl], ids[sl], labels[sl], valid_len=LENS[sl])
    torch.save({"P": tr.student_store.P.clone(), "world": tr.world}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_match_averaged_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert r0["world"] == 2
    assert torch.equal(r0["P"], r1["P"])  # replicas stay identical

    # single process: gradients of the two shards averaged by hand, same optimizer
    cfg_t, cfg_s, t_sd, s_sd, feats, ids, labels = _make()
    ops = RefOps("cpu", lowp=torch.float32)
    tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.05)
    aux = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.05)
    for _ in range(2):
        aux.student_store.P.copy_(tr.student_store.P)
        aux.student_store.refresh_shadow()
        tr.forward_backward(feats[0:2], ids[0:2], labels[0:2])
        aux.forward_backward(feats[2:4], ids[2:4], labels[2:4])
        tr.student_store.G.add_(aux.student_store.G).mul_(0.5)
        tr.optimizer_step()
    a, b = tr.student_store.P, r0["P"]
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 2e-6, rel


def _empty_shard_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_t, cfg_s, t_sd, s_sd, feats, ids, labels = _make()
    tr = DistillationTrainer(RefOps("cpu", lowp=torch.float32), s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.05)
    sl = slice(rank * 2, rank * 2 + 2)
    lab = labels[sl].clone()
    if rank == 1:
        lab[:] = -100                    # this rank's shard holds no label at all
    tr.train_step(feats[sl], ids[sl], lab)
    torch.save({"P": tr.student_store.P.clone(), "step": tr.step_count}, os.path.join(out_dir, f"e{rank}.pt"))
    # second case: NO rank has a label -> every rank skips the step (on the device), together
    before = tr.student_store.P.clone()
    tr.train_step(feats[sl], ids[sl], torch.full_like(lab, -100))
    torch.save({"same": torch.equal(before, tr.student_store.P), "step": tr.step_count}, os.path.join(out_dir, f"f{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rank_without_labels_steps_with_the_others(tmp_path):
    """The optimizer-skip gate is the label count of the whole data-parallel step, not of the rank: a rank whose shard has
    no label applies the reduced gradient like everybody else (replicas stay identical); only a step without any label
    on any rank is skipped."""
    port = _free_port()
    mp.spawn(_empty_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    e0, e1 = torch.load(os.path.join(tmp_path, "e0.pt")), torch.load(os.path.join(tmp_path, "e1.pt"))
    assert e0["step"] == 1 and e1["step"] == 1
    assert torch.equal(e0["P"], e1["P"])
    f0, f1 = torch.load(os.path.join(tmp_path, "f0.pt")), torch.load(os.path.join(tmp_path, "f1.pt"))
    assert f0["same"] and f1["same"] and f0["step"] == 1 and f1["step"] == 1


# ---------------------------------------------------------------------------------------------------------------------
# eval-side exchange: generated ids padded to a common width and concatenated in rank order
def _gather_worker(rank, world, port, out_dir):
    import numpy as np
    from distil_whisper_amd.gather import gather_rows, gather_token_lists, pad_across_processes
    from distil_whisper_amd.longform import LongFormTranscriber
    from distil_whisper_amd.modeling import WhisperFeatureExtractor, WhisperForConditionalGeneration
    from distil_whisper_amd.pseudo_label import PseudoLabeller
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # primitives: ragged widths and ragged row counts
    mine = torch.arange(1, 1 + (3 - rank) * (4 + 2 * rank)).reshape(3 - rank, 4 + 2 * rank)     # rank0 [3,4], rank1 [2,6]
    padded = pad_across_processes(mine, dim=1, pad_index=-7)
    full = gather_rows(mine, pad_index=-7)
    lists = gather_token_lists([[5] * (rank + 1), [], [9, 9, 0]][:3 - rank], 0, "cpu")
    # the two generation front ends, sharded over the ranks and gathered
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, 4)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    ops = RefOps("cpu", lowp=torch.float32)
    model = WhisperForConditionalGeneration(cfg_s, ops=ops, state_dict=s_sd)
    fe = WhisperFeatureExtractor(feature_size=cfg_s.n_mels, ops=ops)
    rng = np.random.default_rng(10)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (200_000, 150_000, 300_000, 100_000, 250_000)]
    spk = [0, 0, 0, 1, 2]
    eos = cfg_s.vocab - 3
    pl = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=5, eos_token_id=eos, use_graphs=False, rank=rank, world=world)
    labels = pl(audios, spk, gather=True)[0]
    lf = LongFormTranscriber(model, fe, batch_size=2, max_new_tokens=4, first_special_id=cfg_s.vocab - 8, use_graphs=False,
                             rank=rank, world=world)
    texts = lf([audios[2], np.concatenate([audios[0], audios[2], audios[4]]), audios[3]], gather=True)
    partial = lf([audios[2], audios[3], audios[4]])
    torch.save({"padded": padded, "full": full, "lists": lists, "labels": labels, "texts": texts, "partial": partial},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_generated_ids_are_gathered_in_rank_order(tmp_path):
    import numpy as np
    from distil_whisper_amd.longform import LongFormTranscriber
    from distil_whisper_amd.modeling import WhisperFeatureExtractor, WhisperForConditionalGeneration
    from distil_whisper_amd.pseudo_label import PseudoLabeller
    port = _free_port()
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    a, b = torch.arange(1, 13).reshape(3, 4), torch.arange(1, 13).reshape(2, 6)
    assert r0["padded"].shape == (3, 6) and torch.equal(r0["padded"][:, :4], a) and bool((r0["padded"][:, 4:] == -7).all())
    assert torch.equal(r1["padded"], b)
    want = torch.cat([torch.cat([a, torch.full((3, 2), -7)], 1), b], 0)
    assert torch.equal(r0["full"], want) and torch.equal(r1["full"], want)
    assert r0["lists"] == r1["lists"] == [[5], [], [9, 9, 0], [5, 5], []]
    # single-process answers
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, 4)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    ops = RefOps("cpu", lowp=torch.float32)
    model = WhisperForConditionalGeneration(cfg_s, ops=ops, state_dict=s_sd)
    fe = WhisperFeatureExtractor(feature_size=cfg_s.n_mels, ops=ops)
    rng = np.random.default_rng(10)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (200_000, 150_000, 300_000, 100_000, 250_000)]
    eos = cfg_s.vocab - 3
    labels = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=5, eos_token_id=eos, use_graphs=False)(audios, [0, 0, 0, 1, 2])[0]
    assert r0["labels"] == labels and r1["labels"] == labels and len(labels) == 4
    lf = LongFormTranscriber(model, fe, batch_size=2, max_new_tokens=4, first_special_id=cfg_s.vocab - 8, use_graphs=False)
    texts = lf([audios[2], np.concatenate([audios[0], audios[2], audios[4]]), audios[3]])
    assert r0["texts"] == texts and r1["texts"] == texts
    part = lf([audios[2], audios[3], audios[4]])
    assert r0["partial"] == part[:2] + [None] and r1["partial"] == [None, None] + part[2:]


def test_bucket_watchdog_names_the_bucket_that_never_completes():
    """distill.BucketWatchdog: buckets that complete are silent; the first one still pending after the deadline produces one
    message naming step, bucket and element range (bench.py ends the rank with it instead of hanging in the next barrier)."""
    import time
    from distil_whisper_amd.distill import BucketWatchdog
    fired = []
    wd = BucketWatchdog(0.3, rank=3, on_timeout=fired.append, poll_s=0.01)
    wd.start()
    t0 = time.monotonic()
    wd.submit("step 7 bucket 0: elements [100, 200) of the flat gradient (1 MiB)", lambda: True)
    wd.submit("step 7 bucket 1: elements [0, 100) of the flat gradient (1 MiB)", lambda: time.monotonic() - t0 > 0.1)   # slow, fine
    wd.submit("step 8 bucket 0: elements [100, 200) of the flat gradient (1 MiB)", lambda: False)                          # wedged
    deadline = time.monotonic() + 5.0
    while not fired and time.monotonic() < deadline:
        time.sleep(0.02)
    wd.stop()
    assert len(fired) == 1
    assert "[rank 3]" in fired[0] and "step 8 bucket 0" in fired[0] and "[100, 200)" in fired[0]
    # a poll that RAISES must not end the thread silently (round-5 advisor finding): it is logged, the bucket counts as
    # pending and the deadline still fires; the first bucket of a run gets the longer grace period (RCCL's lazy set-up)
    fired2 = []
    wd2 = BucketWatchdog(0.2, rank=1, on_timeout=fired2.append, poll_s=0.01, first_grace_s=0.6)

    def boom():
        raise RuntimeError("hipErrorInvalidDevice")
    t1 = time.monotonic()
    wd2.start()
    wd2.submit("step 0 bucket 0: elements [0, 10) of the flat gradient (0 MiB)", boom)
    while not fired2 and time.monotonic() < t1 + 5.0:
        time.sleep(0.02)
    wd2.stop()
    assert len(fired2) == 1 and "step 0 bucket 0" in fired2[0] and wd2.errors > 0
    assert time.monotonic() - t1 >= 0.6          # the first bucket waited for first_grace_s, not timeout_s


def test_grad_reducer_labels_its_buckets_and_counts_steps():
    """The reducer labels every collective it issues (what the watchdog prints); world 1 with always_reduce exercises it."""
    import torch.distributed as dist
    from distil_whisper_amd.distill import GradReducer
    import tempfile
    f = tempfile.NamedTemporaryFile(delete=False)
    dist.init_process_group("gloo", init_method=f"file://{f.name}", rank=0, world_size=1)
    try:
        flat = torch.ones(1000)
        fired = []
        r = GradReducer(flat, bucket_bytes=4 * 300, always_reduce=True, watchdog_s=5.0, on_timeout=fired.append)
        r.ready(600, 1000)
        r.ready(200, 600)
        r.ready(0, 200)
        assert len(r.labels) == 2 and "[600, 1000)" in r.labels[0] and "[200, 600)" in r.labels[1]
        r.wait()
        assert r.step == 1 and r.handles == [] and r.labels == [] and not fired
        assert float(flat.sum()) == 1000.0
        r.watchdog.stop()
    finally:
        dist.destroy_process_group()


def _mode_selection_worker(rank, world, port, out_dir):
    """bench.dp_mode_selection over gloo with per-rank timings injected: rank 0 measures the side-stream step FASTER than the
    single-stream one, rank 1 measures it SLOWER (or fails it, third scenario) -- the decision must be the same on both ranks
    and follow the slowest rank."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeTrainer:
        overlap_teacher = False
        wgrad = False

        def set_overlap_wgrad(self, on):
            self.wgrad = bool(on)

    out = {}
    # (single-stream ms, side-stream ms or an exception) per rank
    scenarios = {"side_wins_on_both": [(100.0, 90.0), (101.0, 95.0)],
                 "side_loses_on_one": [(100.0, 90.0), (100.0, 130.0)],
                 "side_fails_on_one": [(100.0, 90.0), (100.0, RuntimeError("side-stream launch failed"))],
                 "single_slow_on_one": [(100.0, 99.0), (140.0, 120.0)]}
    for name, per_rank in scenarios.items():
        tr = FakeTrainer()
        single, side = per_rank[rank]

        def probe():
            v = side if tr.overlap_teacher else single
            if isinstance(v, Exception):
                raise v
            return v
        sel = bench.dp_mode_selection(tr, None, dist, "cpu", probe=probe)
        out[name] = {"sel": sel, "overlap_teacher": tr.overlap_teacher, "wgrad": tr.wgrad}
    with open(os.path.join(out_dir, f"sel{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_mode_selection_is_the_same_on_every_rank(tmp_path):
    """bench.py's data-parallel ranks start single-stream and take the side streams only if the slowest rank measures them
    faster and they ran on every rank (round-5 review item 5: an N-GPU run must not land on an unmeasured configuration)."""
    import json
    port = _free_port()
    mp.spawn(_mode_selection_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(os.path.join(str(tmp_path), f"sel{r}.json"))) for r in (0, 1))
    assert r0 == r1                                            # same reduced numbers, same decision, on both ranks
    want = {"side_wins_on_both": (True, 101.0, 95.0), "side_loses_on_one": (False, 100.0, 130.0),
            "side_fails_on_one": (False, 100.0, None), "single_slow_on_one": (True, 140.0, 120.0)}
    for name, (use_side, single_max, side_max) in want.items():
        got = r0[name]
        assert got["overlap_teacher"] == got["wgrad"] == use_side, name
        assert got["sel"]["eager_single_stream"] == single_max and got["sel"]["eager_side_streams"] == side_max, (name, got)
        assert got["sel"]["side_streams_ran_on_every_rank"] == (name != "side_fails_on_one")
