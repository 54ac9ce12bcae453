"""Contract benchmark: training audio-seconds/s of the Whisper distillation step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = the reference hot loop (run_distillation.py:1465-1495, 1606-1614) on one batch of synthetic audio that is
already resident in HBM: log-mel front end + teacher forward + student forward/backward + gradient all-reduce (RCCL,
N > 1) + global-norm clip + AdamW.  Workload = BASELINE.json configs[2]/[3]: whisper-large-v3-shaped teacher (32/32)
-> distil-large-v3-shaped student (32 encoder / 2 decoder layers), per-GPU batch 32 x 30 s clips, nothing frozen
("full" mode; `--mode recipe` = the README recipe with --freeze_encoder and a shared encoder).  Weights are seeded
random (no checkpoints exist offline), data is synthetic.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel class (the bf16 MFMA GEMM): its launches are
bracketed by HIP events on the launch stream during one extra instrumented step after the timed region.
`cpu_baseline` times the reference path itself (the `transformers` classes under the reference's train_step,
oracle/reference_cpu_step.py) on a bounded sample on the host cores.  With N > 1 every rank checks after the timed
steps that its parameters are bit-identical to rank 0's (data-parallel replicas must not diverge).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)


def step_flops(d, mode, T=447, rows=None, sq=None):
    """Algorithmic FLOPs per 30 s sample (2 x MAC, causal attention counted full, no recompute): SURVEY.md section 8(d).
    T = decoder positions the step computes (447, or the live ones when the dead tail of the batch is left out);
    rows = average packed rows per sample of the passes that run over live rows only (the frozen teacher's decoder, the
    row-local work of the student's decoder layers, both LM heads), T when nothing is packed; sq = mean squared sequence
    length (ragged-batch self-attention of the packed teacher decoder: sum of len^2 instead of T^2 per sample)."""
    S = 1500
    D, V, M = d.d_model, d.vocab, d.n_mels
    rows = T if rows is None else rows

    def enc(le):
        return 6 * M * D * 3000 + 6 * D * D * 1500 + le * (24 * S * D * D + 4 * S * S * D)

    def dec(ld, packed=False, ragged_attention=False):
        r = rows if packed else T
        if ragged_attention and sq is not None:     # attention over the live rows only (dw_attn_fwd_varlen)
            return ld * (28 * r * D * D + 4 * S * D * D + 4 * sq * D + 4 * rows * S * D)
        return ld * (28 * r * D * D + 4 * S * D * D + 4 * T * T * D + 4 * T * S * D)

    head = 2 * rows * D * V
    return enc, dec, head


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)      # >= 10 warm-up, >= 30 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the whole step from one captured HIP graph (auto: on with one rank; data-parallel "
                         "ranks issue their RCCL buckets from the host and run eagerly)")
    ap.add_argument("--no-ab", action="store_true", help="skip the in-process A/B legs (graph / eager / eager without "
                                                         "side streams, 10 steps each) reported under `ab`")
    ap.add_argument("--dense", action="store_true",
                    help="compute all 447 decoder positions like the reference does; default: the positions behind the last "
                         "labelled one of the batch (labels -100 for L~U{32..224} onwards, BASELINE.md) are left out -- same "
                         "loss and gradients (distill.trim_dead_positions), reported as decoder_positions_computed")
    ap.add_argument("--no-pack", action="store_true",
                    help="leave out only the common dead tail (one length for the batch); default: per-sequence label lengths, "
                         "the teacher decoder, the LM heads and the loss run over the packed live rows (engine.LiveRows)")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch of 30 s clips")
    ap.add_argument("--model", default="large-v3", choices=["tiny.en", "small.en", "large-v3"])
    ap.add_argument("--mode", default="full", choices=["full", "recipe"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="teacher forward and weight-gradient GEMMs on their own HIP streams (round 2's default; since the "
                         "round-3 epilogue work the single-stream step is ~1 %% faster in the same-process A/B, `ab` below)")
    ap.add_argument("--no-overlap", action="store_true", help="everything on the main stream (one rank without either flag: "
                                                               "decided by a short measurement, see mode_selection)")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="with --overlap: weight-gradient GEMMs of the backward stay on the main stream")
    ap.add_argument("--no-pad-teacher-rows", action="store_true",
                    help="teacher decoder GEMMs over exactly B*T rows (default: padded to a multiple of 320 rows)")
    ap.add_argument("--no-teacher-overlap", action="store_true", help="with --overlap: teacher forward stays on the main stream")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the gradient all-reduce: nccl = RCCL over xGMI (default); gloo "
                         "(through the host) exists to exercise the N-rank path on a box with fewer GPUs")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks on one device); a test of the "
                         "launcher, the rank plumbing and the data-parallel step, not a measurement")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="seconds a gradient bucket may stay pending before the rank ends with an error naming it "
                         "(distill.BucketWatchdog; N > 1 only)")
    ap.add_argument("--cu-hog", type=int, default=0,
                    help="emulate a resident communication kernel: keep this many CUs busy on a side stream during every timed "
                         "step (tools/cu_hog.hip; a one-GPU stand-in for RCCL's channels, used with --share-device)")
    ap.add_argument("--cu-hog-ms", type=float, default=400.0, help="milliseconds of CU occupancy queued per step with --cu-hog")
    ap.add_argument("--no-reference-loop", action="store_true",
                    help="skip the `via_reference_loop` leg (the reference's loop body -- model(**batch), its own fp32 softmax / "
                         "KL lines, loss.backward(), clip_grad_norm_, torch.optim.AdamW -- over the drop-in modules)")
    ap.add_argument("--dp-probe", action="store_true",
                    help="run the data-parallel mode selection (dp_mode_selection) also over gloo (with --share-device: exercises "
                         "the selection and the all-reduced decision on a one-GPU box; over nccl it runs by default)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        le0, ld0 = {"tiny.en": (4, 1), "small.en": (12, 4), "large-v3": (32, 2)}[args.model]
        print(json.dumps(cpu_baseline_leg(args.model, le0, ld0, args.mode == "recipe")), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL), same arguments
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_device:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --share-device needs --backend gloo")
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local_rank}, {torch.cuda.device_count()} device(s) visible")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL runs one workgroup (one CU) per channel for as long as a bucket is in flight.  The step needs 5.3 GB per GPU
        # moved (ring all-reduce of 3.0 GB) inside ~250 ms of backward: 21 GB/s -- a handful of channels -- while every CU
        # RCCL holds is one the persistent GEMM grids (256 workgroups, tiles drawn from per-XCD counters) do without:
        # 8 CUs held for the WHOLE step cost +6 % (tools/comm_contention.py), so 8 channels active for ~1/5 of it cost ~1 %.
        # A caller's own setting wins.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "4")
        torch.cuda.set_device(local_rank)
        # a collective that never completes must end the run with a message, not hang it: torch's own watchdog gets the same
        # deadline as the per-bucket one (distill.BucketWatchdog names the bucket; this one covers init and the barriers)
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(60.0, 4 * args.comm_timeout))
        if args.backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"),
                                    timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from distil_whisper_amd.ops_hip import HipOps          # raises if the HIP library or the GPU is missing
    from distil_whisper_amd.distill import DistillationTrainer
    from distil_whisper_amd import student_init as si

    ops = HipOps(dev)
    tdims = si.PRESETS[args.model]
    le, ld = si.STUDENT_LAYERS[args.model]
    t_sd = si.random_state_dict(tdims, seed=0, device=dev)
    s_sd, sdims = si.student_from_teacher(t_sd, tdims, le, ld)
    filt = torch.tensor(si.mel_filter_bank(tdims.n_mels), dtype=torch.float32, device=dev).contiguous()
    recipe = args.mode == "recipe"
    # Side streams (frozen teacher forward, weight-gradient GEMMs): decided by measurement unless told.  One rank: HIP graph
    # on one stream against eager with side streams (mode_selection below).  Data-parallel ranks run the eager step -- the
    # RCCL buckets are issued from the host between the backward's layers -- and START on the configuration every test
    # covers: everything on the main stream + the reducer's communication stream.  Then (dp_mode_selection below) a few
    # untimed steps of that against the same step with the side streams on, max over ranks, and ALL ranks take the faster
    # one (the choice is all-reduced, so no rank can disagree); a failure of the side-stream probe on any rank falls back to
    # the single-stream step.  --overlap / --no-overlap skip the probe.  Over gloo on a shared GPU (--share-device, a
    # plumbing test) the side streams stay off unless --overlap is given: gloo's host round trips serialise them (12x
    # slower, measured in round 3).
    side = args.overlap and not args.no_overlap
    tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, temperature=2.0, kl_weight=1.0, lr=1e-4,
                             weight_decay=0.0, max_grad_norm=1.0, freeze_encoder=recipe, share_encoder=recipe,
                             mel_filters=filt, overlap_teacher=side and not args.no_teacher_overlap,
                             overlap_wgrad=side and not args.no_wgrad_overlap,
                             pad_teacher_rows=not args.no_pad_teacher_rows,
                             comm_watchdog_s=args.comm_timeout if world > 1 else 0.0)
    del t_sd, s_sd
    torch.cuda.empty_cache()

    B, T = args.batch, 447
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    audio = 0.1 * torch.randn(B, 480000, generator=g, device=dev)
    ids = torch.randint(0, 50257, (B, T + 1), generator=g, device=dev)
    ids[:, 0] = tdims.decoder_start_token_id
    lens = torch.randint(32, 225, (B,), generator=g, device=dev)
    dec_in = ids[:, :-1].contiguous()
    labels = ids[:, 1:].clone()
    labels[torch.arange(T, device=dev)[None, :] >= lens[:, None]] = -100
    # host integer, known from the label lengths before a batch goes to the device (collator.report_valid_len); read
    # here once, outside the timed region
    lens_host = [min(T, int(x)) for x in lens.tolist()]
    valid_len = None if args.dense else (max(lens_host) if args.no_pack else lens_host)
    Te = T if valid_len is None else max(lens_host)
    rows_avg = sum(lens_host) / B if isinstance(valid_len, list) and sum(lens_host) < tr.pack_live_rows_below * B * Te else Te
    sq_avg = sum(x * x for x in lens_host) / B if rows_avg != Te else None     # (packed passes: mean squared live length)

    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1)
    if use_graph and world > 1:
        raise SystemExit("bench.py: --graph on needs one rank (data-parallel steps run eagerly)")

    def eager_step(vl=valid_len):
        feats = tr.features(audio)
        return tr.train_step(feats, dec_in, labels, valid_len=vl)

    def graph_step(vl=valid_len):
        return tr.train_step_graphed(audio, dec_in, labels, valid_len=vl)

    one_step = graph_step if use_graph else eager_step

    # label lengths that CHANGE from step to step (an A/B leg): real batches have their own (max, sum) of label lengths, so
    # the captured plans are keyed by quantised sizes (DistillationTrainer.plan_pos_quantum / plan_row_quantum)
    fresh = []
    for i in range(8):
        li = torch.randint(32, 225, (B,), generator=g, device=dev)
        lab_i = ids[:, 1:].clone()
        lab_i[torch.arange(T, device=dev)[None, :] >= li[:, None]] = -100
        fresh.append((lab_i, [min(T, int(x)) for x in li.tolist()]))
    fresh_i = [0]

    def fresh_step():
        lab_i, lens_i = fresh[fresh_i[0] % len(fresh)]
        fresh_i[0] += 1
        return tr.train_step_graphed(audio, dec_in, lab_i, valid_len=lens_i)

    if args.cu_hog > 0:
        # `--cu-hog N`: N workgroups that each hold a CU (64 KiB of LDS: no GEMM workgroup fits beside one) spin on a side
        # stream while the step runs -- what RCCL's channel kernels do to the persistent GEMM grids, on one GPU
        import ctypes
        import subprocess
        tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")
        so = os.path.join(tools, "libcuhog.so")
        if not os.path.exists(so):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                                   os.path.join(tools, "cu_hog.hip"), "-o", so])
        hog = ctypes.CDLL(so)
        hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        hog_stream, hog_sink = torch.cuda.Stream(), torch.zeros(4, device=dev)
        hog_reps = max(1, int(args.cu_hog_ms / 5.0 + 0.5))
        plain_step = one_step

        def one_step():
            for _ in range(hog_reps):
                hog.hog_launch(args.cu_hog, 5000, hog_sink.data_ptr(), ctypes.c_void_p(hog_stream.cuda_stream))
            return plain_step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"engine ready: {args.model} {args.mode} B={B} world={world} graph={use_graph} decoder positions {Te}/{T}, "
        f"{rows_avg:.1f} packed rows per sample; "
        f"warm-up x{args.warmup}")
    selection = None
    if args.graph == "auto" and world == 1 and not (args.overlap or args.no_overlap):
        # Two ways to issue the same step, a few untimed steps of each, the faster one runs the timed region: the whole step
        # replayed from one HIP graph on one stream (no host work at all), or issued eagerly with the frozen teacher and
        # the weight-gradient GEMMs on their own streams (fills the tails of the small decoder GEMMs; needs a host that
        # keeps up with ~7 launches per millisecond).  Which one wins depends on the box's host as well as its GPU.
        def probe(step_fn, pre):
            for _ in range(pre):
                step_fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record()
            for i in range(6):
                step_fn()
                ev[i + 1].record()
            torch.cuda.synchronize()
            return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(6))[2]
        try:
            selection = {"hip_graph_single_stream": probe(graph_step, 4)}
        except Exception as e:      # a box on which the capture fails still gets its measurement from the eager step
            log(f"HIP-graph capture of the step failed ({type(e).__name__}: {e}); eager steps only")
            selection = {"hip_graph_single_stream": float("inf")}
            torch.cuda.synchronize()
        tr.drop_graph()
        torch.cuda.empty_cache()
        tr.overlap_teacher = True
        tr.set_overlap_wgrad(True)
        selection["eager_side_streams"] = probe(eager_step, 2)
        use_graph = selection["hip_graph_single_stream"] <= selection["eager_side_streams"]       # (inf: capture failed)
        if use_graph:
            tr.overlap_teacher = False
            tr.set_overlap_wgrad(False)
        one_step = graph_step if use_graph else eager_step
        selection = {k: (v if v != float("inf") else None) for k, v in selection.items()}
        log("mode selection (median ms of 6 untimed steps): " + json.dumps(selection) +
            f" -> {'hip_graph_single_stream' if use_graph else 'eager_side_streams'}")
    if world > 1 and (args.backend == "nccl" or args.dp_probe) and not (args.overlap or args.no_overlap):
        selection = dp_mode_selection(tr, eager_step, dist, dev)
    if use_graph:
        for _ in range(3):            # two eager steps on the capture stream, then the capture (untimed, before the warm-up)
            one_step()
        torch.cuda.synchronize()
        log("step captured into a HIP graph")
    for i in range(args.warmup):
        losses = one_step()
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss={float(losses[2].item()):.4f}, "
                f"mem={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sync()
    log(f"timing {args.steps} steps")
    torch.cuda.reset_peak_memory_stats()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_ms = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        h0 = time.perf_counter()
        losses = one_step()
        marks[i + 1].record()                      # (main stream, after the optimizer: one event per step)
        host_ms.append((time.perf_counter() - h0) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms = torch.cuda.memory_stats()
    step_stats = {"gpu_ms_median": per_step[len(per_step) // 2], "gpu_ms_p90": per_step[int(0.9 * (len(per_step) - 1))],
                  "gpu_ms_min": per_step[0], "gpu_ms_max": per_step[-1],
                  "host_enqueue_ms_median": sorted(host_ms)[len(host_ms) // 2], "host_enqueue_ms_max": max(host_ms),
                  "reserved_peak_gib": ms.get("reserved_bytes.all.peak", 0) / 2**30,
                  "allocated_peak_gib": ms.get("allocated_bytes.all.peak", 0) / 2**30,
                  "num_alloc_retries": ms.get("num_alloc_retries", 0), "num_device_alloc": ms.get("num_device_alloc", 0),
                  "num_device_free": ms.get("num_device_free", 0)}
    log("per-step: " + json.dumps(step_stats))
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms_per_step = dt / args.steps * 1e3
    log(f"{ms_per_step:.1f} ms/step")
    if world > 1:
        # replicas must stay bit-identical: compare every rank's master weights with rank 0's (outside the timed region)
        P = tr.student_store.P
        ref = P.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([1 if torch.equal(ref, P) else 0], device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        del ref
        if int(same.item()) != 1:
            raise RuntimeError("data-parallel replicas diverged: parameters differ between ranks after the timed steps")
        log("replica check: parameters bit-identical on all ranks")
    value = world * B * 30.0 * args.steps / dt
    loss_val = float(losses[2].item())

    # what runs over the packed live rows: the frozen teacher's decoder incl. its attention (ragged batches), the row-local
    # work of the student's decoder layers (engine.pack_train_layers; their attention keeps the rectangle), both LM heads
    ragged, spack = bool(tr.teacher.varlen_attention), bool(tr.student.pack_train_layers)      # (read now: `tr` is released before the JSON line)

    def sample_flops(Tc, rows=None, sq=None):
        enc, dec, head = step_flops(tdims, args.mode, Tc, rows, sq)
        if recipe:
            return enc(tdims.enc_layers) + (dec(tdims.dec_layers, True, ragged) + head) + 3 * (dec(ld, spack) + head)
        return (enc(tdims.enc_layers) + dec(tdims.dec_layers, True, ragged) + head) + 3 * (enc(le) + dec(ld, spack) + head)
    fl = sample_flops(Te, rows_avg, sq_avg)     # the flops the step EXECUTES (the fraction of peak is priced on these)
    step_tflops = value / 30.0 * fl / 1e12 / world

    ab = None
    if world == 1 and not args.no_ab:
        try:
            ab = ab_legs(tr, eager_step, graph_step, B, dense=None if args.dense else (lambda: graph_step(None)),
                         trimmed=(lambda: graph_step(Te)) if isinstance(valid_len, list) else None,
                         fresh=fresh_step if isinstance(valid_len, list) else None)
        except Exception as e:       # (the legs are extra information: the timed result above is reported regardless)
            ab = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()
        log("A/B (median ms/step over 10 steps, same process): " + json.dumps(ab))
    tr.drop_graph()
    torch.cuda.empty_cache()

    def roofline_leg():
        ops.profile = {}
        overlap, tr.overlap_teacher = tr.overlap_teacher, False   # one kernel at a time for the per-launch events
        wgrad_overlap = tr.student.wgrad_stream is not None
        tr.set_overlap_wgrad(False)
        one_step()
        torch.cuda.synchronize()
        prof = ops.collect_profile()
        ops.profile = None
        tr.overlap_teacher = overlap
        tr.set_overlap_wgrad(wgrad_overlap)
        log("per-kernel-class ms (instrumented step): " + json.dumps(
            {k: {"n": v["n"], "ms": round(v["ms"], 2), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}))
        if prof:
            key = max(prof, key=lambda k: prof[k]["ms"])
            p = prof[key]
            return {"bound": "mfma", "kernel": key, "launches": p["n"],
                        "avg_launch_ms": p["ms"] / p["n"], "flops_per_launch": p["flops"] / p["n"],
                        "achieved": p["flops"] / (p["ms"] * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": p["flops"] / (p["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "traffic": pmc_traffic(key, args, p["n"]),
                        "vendor_library_tflops": vendor_ceiling(),
                        "attention_vendor_tflops": attn_vendor_ceiling(),
                        "all_gemm_ms": sum(v["ms"] for k, v in prof.items() if k.startswith("gemm")),
                        "all_gemm_tflops": sum(v["flops"] for k, v in prof.items() if k.startswith("gemm")) /
                        max(sum(v["ms"] for k, v in prof.items() if k.startswith("gemm")), 1e-9) / 1e9,
                        "attn_ms": sum(v["ms"] for k, v in prof.items() if k.startswith("attn")),
                        "step_ms_instrumented": sum(v["ms"] for v in prof.values())}
        return None

    roofline = None
    if not args.no_roofline:
        try:
            roofline = roofline_leg()
        except Exception as e:         # (extra information: the timed result is reported regardless)
            ops.profile = None
            roofline = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()

    step_mode = ("hip_graph" if use_graph else "eager") + \
        ("_side_streams" if (tr.overlap_teacher or tr.student.wgrad_stream is not None) else "_single_stream")
    red = tr.reducer
    tr_info = {"bucket_mib": (red.bucket_elems * 4 / 2**20) if red is not None else None,
               "buckets": getattr(red, "last_buckets", None), "reduced_mib": getattr(red, "last_bytes", 0) / 2**20 if red is not None else None,
               "streams": ["main"] + (["teacher"] if tr.overlap_teacher else []) +
                          (["weight-gradient"] if tr.student.wgrad_stream is not None else []) +
                          (["communication"] if red is not None and red.active and red.stream is not None else [])}
    via_loop = None
    if world == 1 and not args.no_reference_loop:
        # what a maintainer gets from the two-line import change alone: the drop-in modules under the
        # reference's own loop body, optimizer and loss lines -- timed here so that the cost of staying on them is a number
        del tr
        torch.cuda.empty_cache()
        try:
            via_loop = reference_loop_leg(ops, args, tdims, le, ld, audio, dec_in, labels, lens_host, filt, B, dev)
        except Exception as e:         # (extra information: the timed result is reported regardless)
            via_loop = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()
        log("reference loop over the drop-in modules: " + json.dumps(via_loop))

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline leg (subprocess, bounded to 420 s) ...")
        cpu_baseline = run_cpu_baseline(args.model, args.mode)

    if rank == 0:
        out = {"metric": "training audio-seconds/s (distil-large-v3 student, 30 s clips)" if args.model == "large-v3"
               else f"training audio-seconds/s ({args.model} student, 30 s clips)",
               "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"whisper-{args.model} teacher ({tdims.enc_layers}/{tdims.dec_layers}) -> "
                                      f"{le}/{ld} student KD step, mode={args.mode}",
                          "global_batch": B * world, "per_gpu_batch": B, "clip_seconds": 30, "decoder_len": T,
                          "decoder_positions_computed": Te, "packed_rows_per_sample": rows_avg,
                          "parallelism": f"dp{world}" + (f" ({args.backend}, ranks share cuda:0: plumbing test)" if args.share_device else ""), "mode": args.mode, "includes": "logmel+teacher_fwd+student_fwd_"
                          "bwd+allreduce+clip+adamw", "loss": loss_val},
               "step_tflops_per_gpu": step_tflops, "step_mfma_frac": step_tflops / PEAK_BF16_TFLOPS,
               "flops_per_sample": fl, "flops_per_sample_all_positions": sample_flops(T), "step_mode": step_mode, "cu_hog": args.cu_hog or None,
               "mode_selection": selection, "step_stats": step_stats,
               "collective": None if world == 1 else {
                   "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""), "ranks": world,
                   "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                   "bucket_mib": tr_info["bucket_mib"], "buckets_per_step": tr_info["buckets"],
                   "all_reduced_mib_per_step": tr_info["reduced_mib"], "streams": tr_info["streams"]},
               "kernels_sha16": _kernels_sha16(),
               "ab": ab, "via_reference_loop": via_loop, "roofline": roofline, "cpu_baseline": cpu_baseline}
        if ab and "hip_graph_all_447_decoder_positions" in ab and "error" not in ab:
            # the step exactly as the reference shapes it (every padded decoder position computed), same process
            d447 = ab["hip_graph_all_447_decoder_positions"]
            out["all_447_decoder_positions"] = {"ms_per_step": d447["median_ms"], "value": d447["audio_s_per_s"],
                                                "unit": "audio-s/s", "steps": 10,
                                                "step_mfma_frac": d447["audio_s_per_s"] / 30.0 * sample_flops(T) / 1e12 /
                                                PEAK_BF16_TFLOPS}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def reference_loop_leg(ops, args, tdims, le, ld, audio, dec_in, labels, lens_host, filt, B, dev, warm=2, steps=4):
    """The reference's hot loop over the DROP-IN surface (distil_whisper_amd.modeling), exactly as run_distillation.py drives
    it: `student_model(**batch)`, `teacher_model(**batch)`, the fp32 softmax / log_softmax / kl_divergence lines,
    `loss.backward()`, `clip_grad_norm_`, `torch.optim.AdamW.step()`, `zero_grad()` (1465-1495, 1606-1614).  The loop body is
    the verbatim restatement the parity tests use (oracle/reference_loop.py) -- here it is the CALLER of the product, not
    part of it.  Three legs, median ms per step: the batch as the reference's collator emits it; the same with the label
    lengths in the batch (`valid_len`, what the drop-in collator adds: dead decoder positions left out); and that plus the
    one-call fused KD loss (modeling.fused_distillation_loss) instead of the reference's softmax lines."""
    from distil_whisper_amd import modeling as M
    from distil_whisper_amd import student_init as si
    from oracle.reference_loop import ReferenceLoop
    recipe = args.mode == "recipe"
    t_sd = si.random_state_dict(tdims, seed=0, device=dev)
    s_sd, sdims = si.student_from_teacher(t_sd, tdims, le, ld)
    student = M.WhisperForConditionalGeneration(sdims, ops=ops, state_dict=s_sd)
    teacher = M.WhisperForConditionalGeneration(tdims, ops=ops, state_dict=t_sd, dtype=torch.bfloat16)
    del t_sd, s_sd
    if recipe:
        student.freeze_encoder()
    out = {"steps": steps, "warmup": warm, "optimizer": "torch.optim.AdamW (two groups) + clip_grad_norm_", "unit": "ms/step"}

    def leg(with_len, fused, fused_opt=False, skip_dead=False):
        import functools
        # (DW_SKIP_DEAD_POSITIONS=1 in the environment of the unedited script sets this class attribute at import)
        student.skip_dead_positions = teacher.skip_dead_positions = bool(skip_dead)
        from distil_whisper_amd.optim import FusedAdamW
        loop = ReferenceLoop(student, teacher, M.BaseModelOutput, share_hidden_states=recipe, teacher_dtype=torch.bfloat16,
                             fused_loss=M.fused_distillation_loss if fused else None,
                             optimizer_cls=functools.partial(FusedAdamW, model=student) if fused_opt else None)

        def one():
            feats = ops.logmel(audio, filt)                      # (the front end is part of the step, as in the main run)
            batch = {"input_features": feats, "decoder_input_ids": dec_in, "labels": labels}
            if with_len:
                batch["valid_len"] = lens_host
            return loop.training_iteration(batch, temperature=2.0)
        for _ in range(warm):
            m, _ = one()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            m, _ = one()
            marks[i + 1].record()
        torch.cuda.synchronize()
        t = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        return {"median_ms": t[len(t) // 2], "audio_s_per_s": B * 30e3 / t[len(t) // 2], "loss": float(m["loss"].item())}
    from distil_whisper_amd import lazy_logits
    before = dict(lazy_logits.STATS)
    out["verbatim"] = leg(False, False)
    # (the loop's own softmax / log_softmax / KLDivLoss lines run over the drop-in's lazy `.logits`: answered by the fused loss
    # kernel, no fp32 [B, T, V] temporaries -- distil_whisper_amd/lazy_logits.py; the counters say whether that happened)
    out["verbatim"]["lazy_logits"] = {k: lazy_logits.STATS[k] - before[k] for k in before}
    # the unedited loop again with DW_SKIP_DEAD_POSITIONS=1: the models read the label lengths back from `labels` themselves
    # (behind the encoder's launches) and leave the dead decoder positions out -- no collator change, no `valid_len` in the batch
    out["verbatim_env_DW_SKIP_DEAD_POSITIONS"] = leg(False, False, skip_dead=True)
    # the same loop body with distil_whisper_amd.optim.FusedAdamW in place of torch.optim.AdamW and its clip_grad_norm_ in
    # place of accelerator.clip_grad_norm_ (two changed lines of the script; same parameter groups, same LambdaLR)
    out["verbatim_with_fused_optimizer"] = leg(False, False, True)
    if not recipe:       # (the shared-encoder teacher call of the reference passes labels only: no lengths reach it)
        out["with_valid_len"] = leg(True, False)
        out["with_valid_len_and_fused_kd_loss"] = leg(True, True)
    # (shared encoder: the one-call loss takes the student's rows out of the teacher's full logits)
    out["with_valid_len_fused_kd_loss_and_fused_optimizer"] = leg(True, True, True)
    return out


def dp_mode_selection(tr, eager_step, dist, dev, pre=2, n=5, probe=None):
    """Data-parallel ranks choose between two ways to issue the same eager step -- everything on the main stream (plus the
    reducer's communication stream), or with the frozen teacher forward and the weight-gradient GEMMs on their own streams
    -- by measurement, a few untimed steps of each WITH the bucketed all-reduce in them.  Per leg the slowest rank counts
    (all-reduce MAX of the medians), the side-stream leg is only eligible if it ran on EVERY rank (all-reduce MIN of an ok
    flag), and because every rank reads the same reduced numbers every rank takes the same decision.  The run starts on the
    single-stream step: the configuration the test-suite executes on a GPU (tests/test_dp_gpu.py) -- an 8-GPU run can not
    land on a configuration nobody has run without having measured it against that one first.
    `probe`: the timing function (median ms of n steps after `pre` untimed ones); the default brackets the steps with HIP
    events; tests/test_dp_gloo.py injects per-rank numbers to check the agreement logic on CPU over gloo."""
    def hip_probe():
        for _ in range(pre):
            eager_step()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            eager_step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]

    probe = probe or hip_probe
    tr.overlap_teacher = False
    tr.set_overlap_wgrad(False)
    single = probe()
    ok, sidet = 1.0, float("inf")
    try:
        tr.overlap_teacher = True
        tr.set_overlap_wgrad(True)
        sidet = probe()
    except Exception as e:      # noqa: BLE001 -- whatever the side-stream step raises, the run continues on the other one
        ok = 0.0
        log(f"side-stream probe failed on this rank ({type(e).__name__}: {e}); single-stream step")
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    t = torch.tensor([single, sidet if ok else 1e30], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    okt = torch.tensor([ok], device=dev, dtype=torch.float64)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    single_max, side_max, all_ok = float(t[0]), float(t[1]), bool(okt.item() == 1.0)
    use_side = all_ok and side_max < single_max
    tr.overlap_teacher = use_side
    tr.set_overlap_wgrad(use_side)
    sel = {"eager_single_stream": single_max, "eager_side_streams": side_max if all_ok else None,
           "side_streams_ran_on_every_rank": all_ok, "statistic": f"max over ranks of the median of {n} untimed steps, ms"}
    log("data-parallel mode selection: " + json.dumps(sel) + f" -> {'eager_side_streams' if use_side else 'eager_single_stream'}")
    return sel


def spawn_ranks(n):
    """Re-exec this script under torch.distributed.run with one rank per GPU of this node (what the driver does for
    N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`)."""
    import socket
    import subprocess
    if torch.cuda.device_count() < n and "--share-device" not in sys.argv:
        print(f"bench.py: --gpus {n} requested but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def ab_legs(tr, eager_step, graph_step, B, steps=10, warm=3, dense=None, trimmed=None, fresh=None):
    """The same process, the same weights and inputs, seconds apart: median GPU ms per step (HIP events around each
    step on the main stream) of the ways to issue the step; `dense`: the graphed step over all 447 decoder positions
    (what the reference computes) when the main run leaves the dead ones out."""
    def leg(step_fn, pre=0):
        for _ in range(pre + warm):
            step_fn()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            step_fn()
            marks[i + 1].record()
        torch.cuda.synchronize()
        t = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        return {"median_ms": t[len(t) // 2], "min_ms": t[0], "max_ms": t[-1], "audio_s_per_s": B * 30e3 / t[len(t) // 2]}

    out = {}
    ov_t, ov_w = tr.overlap_teacher, tr.student.wgrad_stream is not None
    tr.drop_graph()
    torch.cuda.empty_cache()
    out["hip_graph_side_streams" if (ov_t or ov_w) else "hip_graph_single_stream"] = leg(graph_step, pre=3)
    if dense is not None:
        tr.drop_graph()
        torch.cuda.empty_cache()
        out["hip_graph_all_447_decoder_positions"] = leg(dense, pre=3)
    if trimmed is not None:
        tr.drop_graph()
        torch.cuda.empty_cache()
        out["hip_graph_common_dead_tail_only"] = leg(trimmed, pre=3)
    if fresh is not None:
        # eight batches with their own label lengths in turn: every plan of the cycle is captured in the untimed steps
        # (8 batches x (2 eager + 1 capture)), the timed ones replay whichever plan the batch's quantised sizes select
        tr.drop_graph()
        torch.cuda.empty_cache()
        out["hip_graph_label_lengths_change_every_step"] = dict(leg(fresh, pre=21), plans=len(tr._graphs))
    tr.drop_graph()
    torch.cuda.empty_cache()
    out["eager_side_streams" if (ov_t or ov_w) else "eager_single_stream"] = leg(eager_step)
    tr.overlap_teacher = not (ov_t or ov_w)
    tr.set_overlap_wgrad(not (ov_t or ov_w))
    flipped = "single_stream" if (ov_t or ov_w) else "side_streams"
    out["eager_" + flipped] = leg(eager_step)
    torch.cuda.empty_cache()
    out["hip_graph_" + flipped] = leg(graph_step, pre=3)
    tr.drop_graph()
    torch.cuda.empty_cache()
    tr.overlap_teacher = ov_t
    tr.set_overlap_wgrad(ov_w)
    return out


def _kernels_sha16():
    from distil_whisper_amd.build import kernels_sha16
    return kernels_sha16()


def vendor_ceiling():
    """Second ceiling next to the 2.5 PFLOP/s peak: what the vendor GEMM library reaches on the step's
    row-major shapes on an MI355X of this pool (committed measurement, tools/hipblaslt_probe.py); None if not committed.
    (The vendor's side of the calibration does not depend on this library's kernels; the file says which kernel sources
    its `this_library_tflops` column was measured with.)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "vendor_gemm_ceiling.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("row_major_forward_median_tflops")


def attn_vendor_ceiling():
    """What torch's scaled_dot_product_attention reaches on the encoder self-attention shape on a box of this pool
    (tools/attn_vendor_probe.py, committed): {"fwd_tflops", "bwd_tflops"} of its best backend, or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "attn_vendor_ceiling.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        cls = json.load(f).get("classes", [])
    for c in cls:
        if c.get("Lq") == 1500 and c.get("Lk") == 1500:
            ok = [v for v in c.get("vendor", {}).values() if "error" not in v]
            if ok:
                return {"fwd_tflops": max(v["fwd_tflops"] for v in ok), "bwd_tflops": max(v["bwd_tflops"] for v in ok),
                        "this_library_fwd_tflops": c["mine"]["fwd_tflops"], "this_library_bwd_tflops": c["mine"]["bwd_tflops"]}
    return None


def pmc_traffic(key, args, launches_counted=None):
    """L2-fabric-side bytes per launch of the reported kernel class, from the committed PMC passes of this same
    workload (tools/pmc_traffic.py; FETCH_SIZE and WRITE_SIZE cannot be collected inside a timed run).  None -- with the
    reason logged -- when the committed summary is for another workload, does not hold the class, or was measured with
    other kernel sources than the ones this run built (kernels_sha16)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or args.model != "large-v3" or args.mode != "full" or args.batch != 32:
        return None
    with open(path) as f:
        j = json.load(f)
    sha = _kernels_sha16()
    if j.get("kernels_sha16") != sha:
        log(f"roofline.traffic: profiles/pmc_traffic.json was measured with kernel sources {j.get('kernels_sha16')}, "
            f"this run has {sha}: not quoted (re-take it with tools/profile_round.sh)")
        return None
    c = j.get("classes", {}).get(key)
    if c is None:
        log(f"roofline.traffic: profiles/pmc_traffic.json holds no class {key}: not quoted")
        return None
    # the PMC passes must have seen the launches this run's instrumented step counted for the class: a kernel symbol missing
    # from tools/pmc_traffic.py's table (round 4: gemm_wp16_kernel) silently drops launches from the average otherwise
    per_step = c.get("launches_per_step", c["launches"] / max(1, j.get("steps_profiled", 2)))
    if launches_counted is not None and abs(per_step - launches_counted) > 0.02 * launches_counted:
        log(f"roofline.traffic: profiles/pmc_traffic.json covers {per_step:.0f} launches per step of {key}, the instrumented "
            f"step counted {launches_counted}: not quoted (a kernel symbol is missing from tools/pmc_traffic.py CLASS_OF?)")
        return None
    return c["traffic_bytes_per_launch"]


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may actually use (affinity mask and cgroup quota), not the host's total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def run_cpu_baseline(model, mode, limit_s=420):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", model, "--mode", mode]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "audio-s/s", "cores": usable_cores(), "kind": "port",
                "sample": f"failed: {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "audio-s/s", "cores": usable_cores(), "kind": "port",
                "sample": f"1 step, batch 1 x 30 s, {model} dims did not finish within {limit_s} s"}


def cpu_baseline_leg(model, le, ld, recipe):
    """The reference path on the host cores of this box: the `transformers` Whisper classes +
    feature extractor driven by the reference's train_step / AdamW groups / clip (oracle/reference_cpu_step.py), fp32,
    batch 1 x 30 s of the same model configuration, log-mel included, 1 warm-up + up to 3 timed steps within a time
    budget.  Falls back to the oracle port (kind "port") only if `transformers` cannot be imported."""
    from oracle import whisper_oracle as wo
    cfg_t = wo.CONFIGS[model]
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    try:
        from oracle.reference_cpu_step import timed_reference_steps
        r = timed_reference_steps(cfg_t, le, ld, batch=1, recipe=recipe, warmup=1, steps=3, budget_s=200.0,
                                  threads=ncores)
        return {"value": 30.0 * r["batch"] / r["seconds_per_step"], "unit": "audio-s/s", "cores": ncores,
                "kind": "reference",
                "sample": f"transformers WhisperForConditionalGeneration + WhisperFeatureExtractor, reference "
                          f"train_step + clip + AdamW, {model} dims, fp32, batch {r['batch']} x 30 s, log-mel included, "
                          f"{r['warmup']} warm-up + {r['steps']} timed steps ({r['seconds_per_step']:.1f} s/step)"}
    except ImportError as e:
        note = f"transformers unavailable ({e}); oracle port instead"
    t_sd = wo.init_state_dict(cfg_t, 0)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, le, ld)
    b = wo.synthetic_batch(cfg_t, 1, seed=1234, with_audio=True)
    params = {}
    for k, v in s_sd.items():
        rg = k != "model.encoder.embed_positions.weight" and not (recipe and k.startswith("model.encoder."))
        params[k] = v.requires_grad_(rg)
    t0 = time.perf_counter()
    feats = torch.tensor(wo.logmel(b["audio"], cfg_t.n_mels))
    batch = {"input_features": feats, "decoder_input_ids": b["decoder_input_ids"], "labels": b["labels"]}
    loss, *_ = wo.train_step(params, cfg_s, t_sd, cfg_t, batch, 2.0, 1.0, recipe)
    loss.backward()
    grads = {k: p.grad for k, p in params.items() if p.grad is not None}
    with torch.no_grad():
        wo.clip_and_adamw({k: p.detach() for k, p in params.items()}, grads, {}, step=1)
    dt = time.perf_counter() - t0
    return {"value": 30.0 / dt, "unit": "audio-s/s", "cores": ncores, "kind": "port",
            "sample": f"{note}: 1 step, batch 1 x 30 s, {model} dims, fp32 torch CPU ({dt:.1f} s; log-mel included)"}


if __name__ == "__main__":
    main()
