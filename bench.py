#!/usr/bin/env python3
"""bench.py — M reads/s of the fused per-read pass (add barcode + demultiplex + trim + mask by quality).

Workload (BASELINE.json configs[3], the one the metric is quoted on): 2x150 bp paired clusters, 96 dual-index
8+8 barcodes (`i7+i5`, 17 chars), <=1 mismatch, min_baseq 20.  configs[3] is 500 M clusters read-sharded over
8 GPUs; one GPU's shard (62.5 M clusters, 57.8 GB of device-resident SoA buffers) is the per-GPU work at every
N (weak scaling), so N=8 is exactly configs[3].  A "step" = one pass of the hot path over the rank's whole
shard + the count reduce (RCCL all-reduce of u64[S+3] when N > 1).  A "read" is one cluster, as the reference's
own `total_reads` counter counts them (src/fasta_demultiplex.rs:169).

One JSON line on stdout (rank 0).  Inputs are resident in HBM before the timed region; the kernel duration for
the roofline object comes from HIP events recorded on the stream the kernel runs on.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

L_READ = 150
S_SAMPLES = 96
L_BC = 17
MIN_BASEQ = 20
# algorithmic bytes per cluster: read 2*(2*150) + 17, write 2*(150 + 2) + 4
BYTES_PER_PAIR = 2 * (2 * L_READ) + L_BC + 2 * (L_READ + 2) + 4      # 925
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=62_500_000, help="clusters per GPU (default: configs[3] / 8)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="clusters timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=1,
                    help="threads for the CPU baseline (the reference's loops are single-threaded, so 1 is the faithful number; "
                         "more threads split the sample into slices)")
    ap.add_argument("--gen-chunk", type=int, default=2_000_000)
    ap.add_argument("--arena", type=int, default=-1,
                    help="carve every matrix of the shard from ONE allocation (made first), 4 KiB-aligned and staggered by this many bytes; "
                         "-1 = one torch allocation per matrix")
    return ap.parse_args()


def gen_shard(torch, dev, n, table_np, seed, chunk, into=None):
    """Synthetic shard on the device (SURVEY.md §8d cfg 4 distributions), generated chunk by chunk.
    `into` = (seq[2], qual[2], bc) preallocated uint8 tensors to fill instead of allocating."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    u8 = torch.uint8
    bases = torch.tensor(list(b"ACGT"), dtype=u8, device=dev)
    alphabet = torch.tensor(list(b"ACGTN"), dtype=u8, device=dev)
    table = torch.tensor(table_np, dtype=u8, device=dev)
    mu = 36.0 - 16.0 * (torch.arange(L_READ, device=dev, dtype=torch.float32) / (L_READ - 1)) ** 2
    if into is not None:
        seq, qual, bc = into
    else:
        seq = [torch.empty((n, L_READ), dtype=u8, device=dev) for _ in range(2)]
        qual = [torch.empty((n, L_READ), dtype=u8, device=dev) for _ in range(2)]
        bc = torch.empty((n, L_BC), dtype=u8, device=dev)
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        for mi in range(2):
            idx = torch.randint(0, 4, (m, L_READ), generator=g, device=dev)
            s = bases[idx]
            s[torch.rand((m, L_READ), generator=g, device=dev) < 0.005] = ord("N")
            seq[mi][r0:r0 + m] = s
            q = torch.randn((m, L_READ), generator=g, device=dev) * 6.0 + mu
            qual[mi][r0:r0 + m] = (q.round_().clamp_(2, 40) + 33).to(u8)
            del idx, s, q
        truth = torch.randint(0, S_SAMPLES, (m,), generator=g, device=dev)
        b = table[truth].clone()
        rows = torch.arange(m, device=dev)
        for lo in (0, 9):                      # error mix per half: 85 % exact, 10 % one substitution, 5 % random 8-mer
            u = torch.rand((m,), generator=g, device=dev)
            sub = (u >= 0.85) & (u < 0.95)
            pos = lo + torch.randint(0, 8, (m,), generator=g, device=dev)
            ai = torch.randint(0, 5, (m,), generator=g, device=dev)
            cur = b[rows, pos]
            ai = torch.where(alphabet[ai] == cur, (ai + 1) % 5, ai)
            b[rows[sub], pos[sub]] = alphabet[ai[sub]]
            rnd = u >= 0.95
            nr = int(rnd.sum())
            if nr:
                b[rnd, lo:lo + 8] = bases[torch.randint(0, 4, (nr, 8), generator=g, device=dev)]
        bc[r0:r0 + m] = b
    return seq, qual, bc


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON).  Everything else that writes to fd 1 — RCCL's version banner comes out
    # of C stdio at exit, torch warnings, ... — is sent to stderr; the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import seqkit_amd
    from seqkit_amd import shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to time)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1 or os.environ.get("SK_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)  # nccl == RCCL on ROCm (xGMI links)
            backend = "nccl"
        except Exception as e:                              # keep the bench alive if RCCL cannot come up: counts via gloo
            sys.stderr.write(f"[bench] RCCL init failed ({e}); falling back to gloo for the count reduce\n")
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo", rank=rank, world_size=world)
            backend = "gloo"

    ctx = seqkit_amd.Context(local_rank)                    # raises if libseqkit_hip.so is missing
    table = synth.make_sheet(S_SAMPLES, 8, dual=True, seed=4)
    ctx.set_barcodes(table, 1)

    n = args.pairs
    if args.arena >= 0:
        sizes = [n * L_READ] * 4 + [n * L_BC] + [n * L_READ] * 2 + [n * 2] * 2 + [n * 4]
        step = [(sz + args.arena + 4095) // 4096 * 4096 for sz in sizes]
        arena = torch.empty(sum(step) + 4096, dtype=torch.uint8, device=dev)
        offs = [sum(step[:i]) for i in range(len(step))]
        cut = [arena[o:o + sz] for o, sz in zip(offs, sizes)]
        seq = [cut[0].view(n, L_READ), cut[2].view(n, L_READ)]
        qual = [cut[1].view(n, L_READ), cut[3].view(n, L_READ)]
        bc = cut[4].view(n, L_BC)
        out_seq = [cut[5].view(n, L_READ), cut[6].view(n, L_READ)]
        lowest_k = [cut[7].view(torch.int16), cut[8].view(torch.int16)]
        assign = cut[9].view(torch.int32)
        gen_shard(torch, dev, n, table, seed=4000 + rank, chunk=args.gen_chunk, into=(seq, qual, bc))
    else:
        seq, qual, bc = gen_shard(torch, dev, n, table, seed=4000 + rank, chunk=args.gen_chunk)
        out_seq = [torch.empty_like(seq[0]) for _ in range(2)]
        lowest_k = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]     # raw u16 storage
        assign = torch.empty((n,), dtype=torch.int32, device=dev)
    counts = torch.zeros((S_SAMPLES + 3,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    # everything below runs on the ctx's own HIP stream (torch sees it as an external stream)
    stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0,
              "out_seq": out_seq[i].data_ptr(), "lowest_k": lowest_k[i].data_ptr()} for i in range(2)]

    def step(ev=None):
        with torch.cuda.stream(stream):
            counts.zero_()
            if ev is not None:
                ev[0].record(stream)
            ctx.fused_pass_dev(n, L_READ, MIN_BASEQ, mates, bc=bc.data_ptr(), bc_stride=L_BC,
                               assign=assign.data_ptr(), counts=counts.data_ptr())
            if ev is not None:
                ev[1].record(stream)
            if backend == "gloo":                            # host round trip (only when RCCL is unavailable)
                h = counts.cpu()
                shard.reduce_counts(h)
                counts.copy_(h)
            else:
                shard.reduce_counts(counts)                  # the path's only cross-shard state (no-op at N=1)

    def fence():
        if backend is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0
    if backend is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(args.steps, 1)

    # ---- size-independent checks on the full shard + bit-exact parity on a sample (oracle = checker only) -----
    total_counts = counts.cpu().numpy().astype(np.uint64)
    S = S_SAMPLES
    assert int(total_counts[S]) == n * world, (int(total_counts[S]), n * world)
    assert int(total_counts[:S].sum()) == int(total_counts[S + 1])
    cpu_baseline = None
    parity = None
    if rank == 0 and args.cpu_sample > 0:
        from oracle import oracle as orc
        ns = min(args.cpu_sample, n)
        h_bc = bc[:ns].cpu().numpy()
        h_seq = [seq[i][:ns].cpu().numpy() for i in range(2)]
        h_qual = [qual[i][:ns].cpu().numpy() for i in range(2)]
        nthr = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        cuts = [ns * k // nthr for k in range(nthr + 1)]

        def cpu_slice(k):
            lo, hi = cuts[k], cuts[k + 1]
            return (orc.demux_batch(table, h_bc[lo:hi], 1)[0],
                    [orc.trim_batch(h_qual[i][lo:hi], None, MIN_BASEQ) for i in range(2)],
                    [orc.mask_batch(h_seq[i][lo:hi], h_qual[i][lo:hi], None, MIN_BASEQ) for i in range(2)])

        t1 = time.perf_counter()
        if nthr == 1:
            parts = [cpu_slice(0)]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nthr) as ex:                     # ctypes releases the GIL around the C calls
                parts = list(ex.map(cpu_slice, range(nthr)))
        cpu_s = time.perf_counter() - t1
        e_assign = np.concatenate([p[0] for p in parts])
        e_k = [np.concatenate([p[1][i] for p in parts]) for i in range(2)]
        e_m = [np.concatenate([p[2][i] for p in parts]) for i in range(2)]
        ok = np.array_equal(assign[:ns].cpu().numpy(), e_assign)
        for i in range(2):
            ok = ok and np.array_equal(lowest_k[i][:ns].cpu().numpy().view(np.uint16), e_k[i])
            ok = ok and np.array_equal(out_seq[i][:ns].cpu().numpy(), e_m[i])
        parity = bool(ok)
        cpu_baseline = {"value": round(ns / cpu_s / 1e6, 4), "unit": "M reads/s", "cores": nthr, "kind": "port",
                        "sample": f"first {ns} clusters of rank 0's shard, same fused work (demultiplex + 2x trim + 2x mask), "
                                  f"C restatement of the reference loops over the packed SoA batch, {nthr} thread(s) "
                                  f"({os.cpu_count()} host cores present; the reference's commands are single-threaded loops); "
                                  "not the Rust binary"}
        if not ok:
            raise SystemExit("PARITY FAILURE: GPU outputs differ from the oracle on the sampled clusters")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world / (elapsed / args.steps) / 1e6
        achieved = BYTES_PER_PAIR * n / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("pairs") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "M reads/s demultiplex (150bp, 96 barcodes) at 1/8 GPUs; % HBM roofline",
            "value": round(value, 3), "unit": "M reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "fasta add barcode + demultiplex fused with trim+mask by quality: "
                                   f"{n} clusters/GPU x 2x150bp paired, 96 dual-index 8+8 barcodes (17 chars), <=1 mismatch, "
                                   "min_baseq 20 (= BASELINE configs[3], 500M clusters read-sharded over 8 GPUs)",
                       "clusters_per_gpu": n, "read_len": L_READ, "barcodes": S_SAMPLES, "barcode_len": L_BC,
                       "min_baseq": MIN_BASEQ, "read_unit": "cluster (as the reference's total_reads counts)",
                       "count_reduce": ("RCCL all-reduce u64[99] per step" if backend == "nccl" else "gloo all-reduce u64[99] per step") if backend else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "sk::tile_pass_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_cluster": BYTES_PER_PAIR,
                         "read_frac": round((617 * n / (kern_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 4)},
            "cpu_baseline": cpu_baseline,
            "parity_sample_ok": parity,
            "identified_frac": round(float(total_counts[S + 1]) / float(total_counts[S]), 4),
        }
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    ctx.close()
    if backend is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
