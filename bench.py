#!/usr/bin/env python3
"""bench.py -- loci/sec of the MCMC haplotype assembler hot path on MI355X.

A "step" is one pass of the hot path (DenovoMCMC.fit for every unit of the batch, burn-in, posterior
summary) over one batch of synthetic loci already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1]: 10k synthetic tetraploid loci, 8 SNVs, 200 reads, 1000 MCMC steps (2 chains, burn 500).
With --gpus N every rank runs the same number of loci (weak scaling; loci are independent, so there is
no data-path collective: only the timing barrier / max-over-ranks).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=10000, help="loci per GPU")
    ap.add_argument("--mcmc-steps", type=int, default=1000)
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--burn", type=int, default=500)
    ap.add_argument("--reads", type=int, default=200)
    ap.add_argument("--snvs", type=int, default=8)
    ap.add_argument("--ploidy", type=int, default=4)
    ap.add_argument("--no-cache", action="store_true", help="disable the per-chain llk cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units for the CPU baseline (0 = auto)")
    return ap.parse_args()


def usable_cores():
    """Threads the CPU baseline may really use: the affinity mask, capped by the cgroup CPU quota when there is one
    (the GPU boxes expose 256 hardware threads but grant a 16-CPU quota: more threads than that only spin)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def call_concordance(args, batch, n=64):
    """Checker leg (rank 0, after the timed region): the first n loci of the batch through the oracle on the same
    Philox streams (llk cache off); the posterior mode genotype and its probability must be those the device-side
    summary of the timed passes reported.  Returns (fraction of equal mode genotypes, max |delta GPM|)."""
    from oracle import binding as orc
    from mchap_amd import GenotypeMultiTrace
    from mchap_amd.assemble import break_table, unpack_trace
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    n = min(n, args.loci)
    reads, _, _ = synth_units(n, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=0)
    cfg = orc.make_cfg(args.ploidy, args.mcmc_steps, args.chains, None, (1.0,), llk_cache_threshold=-1, seed=42,
                       rng_kind=orc.RNG_PHILOX, break_table=break_table(args.snvs, 1.0, 3.0))
    g, l, code, _ = orc.denovo_fit_batch(cfg, reads, [2] * args.snvs, n_threads=0, keep_traces=True)
    assert code == 0
    post = batch.posterior_host()
    fixed = batch.d_fixed.cpu().numpy().reshape(args.loci, args.snvs)
    same, dmax = 0, 0.0
    for u in range(n):
        ref = GenotypeMultiTrace._from_sorted(sort_haplotypes(g[u]), l[u]).burn(args.burn).posterior().mode_genotype_support().mode_genotype()
        mine = unpack_trace(post["mode_words"][u][None], fixed[u], 2)[0]
        same += int(np.array_equal(mine, ref[0]))
        dmax = max(dmax, abs(float(post["stats"][u][1]) - float(ref[1])))
    return same / n, dmax


def cpu_baseline(args, cores, quota=None):
    """The oracle (C restatement incl. the reference's llk trie cache at its default threshold 100) timed on
    the host cores over a bounded sample of the same workload."""
    from oracle import binding as orc
    from mchap_amd.assemble import break_table
    from mchap_amd.synth import synth_units

    n = args.cpu_sample or min(4000, 160 * cores)  # about 10-30 s of CPU work in total
    reads, _, _ = synth_units(n, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=0)
    cfg = orc.make_cfg(args.ploidy, args.mcmc_steps, args.chains, None, (1.0,), llk_cache_threshold=100, seed=42,
                       rng_kind=orc.RNG_PHILOX, break_table=break_table(args.snvs, 1.0, 3.0))
    orc.denovo_fit_batch(cfg, reads[: min(n, cores)], [2] * args.snvs, n_threads=cores, keep_traces=False)  # warm
    t = time.perf_counter()
    _, _, code, st = orc.denovo_fit_batch(cfg, reads, [2] * args.snvs, n_threads=cores, keep_traces=False)
    dt = time.perf_counter() - t
    assert code == 0
    n1 = min(n, 256)  # one thread, ~2 s
    t = time.perf_counter()
    orc.denovo_fit_batch(cfg, reads[:n1], [2] * args.snvs, n_threads=1, keep_traces=False)
    dt1 = time.perf_counter() - t
    return {
        "value": n / dt, "unit": "loci/s", "cores": cores, "kind": "port", "value_1_thread": n1 / dt1,
        "sample": "%d loci of the same workload, oracle/mchap_oracle.c with llk cache threshold 100, OpenMP over loci "
                  "(%d threads = affinity capped by the cgroup CPU quota %s), %.2f s wall"
                  % (n, cores, "none" if quota is None else "%.1f" % quota, dt),
        "llk_evals_per_locus": st.llk_evals / n, "llk_cache_hits_per_locus": st.llk_cache_hits / n,
    }


def pmc_traffic(args):
    """HBM bytes per sampler launch from the rocprofv3 PMC passes (profiles/*_pmc_traffic.json, produced by
    tools/profile_round.sh + tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE converted with factors calibrated in the
    sampler's access width, MI355X_MICROARCH.md "HBM"); the newest summary that matches this workload, else None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("loci") == args.loci and t.get("mcmc_steps") == args.mcmc_steps and t.get("chains") == args.chains:
                return t["hbm_bytes_per_launch"]
        except Exception:
            pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist

        # RCCL over xGMI; MCHAP_BENCH_BACKEND=gloo lets the N > 1 path be exercised by several ranks on ONE GPU
        # (RCCL refuses two ranks on the same device), which is how it is tested on the 1-GPU boxes
        dist.init_process_group(backend=os.environ.get("MCHAP_BENCH_BACKEND", "nccl"))

    from mchap_amd import DenovoMCMC
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    U = args.loci
    first = rank * U  # rank r owns loci [r*U, (r+1)*U): contiguous shard, RNG keyed by global locus id
    reads, _, _ = synth_units(U, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=first)
    model = DenovoMCMC(ploidy=args.ploidy, n_alleles=[2] * args.snvs, steps=args.mcmc_steps, chains=args.chains,
                       random_seed=42, llk_cache_threshold=-1 if args.no_cache else 100)
    batch = DenovoDeviceBatch(model, reads, first_stream=first)
    del reads

    from mchap_amd import _lib

    L = _lib.lib()
    L.mchap_set_profiling(1)  # HIP events on the launch stream right around the sampler kernel
    kernel_ms = []

    def one_pass(events=None):
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        batch.run()
        if events is not None:
            e1.record()
            events.append((e0, e1))
        batch.posterior(args.burn)
        if events is not None:
            kernel_ms.append(L.mchap_last_sampler_ms())  # waits for that launch only

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(events)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    span_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))  # prepare pass + sampler + memsets
    kern_ms = float(np.mean(kernel_ms))
    kern_name = L.mchap_last_sampler_name().decode()

    status = batch.d_status.cpu().numpy()
    if (status > 1).any() or (status < 0).any():
        raise SystemExit("sampler reported errors: %s" % np.unique(status))

    if rank == 0:
        total_units = U * world * args.steps
        value = total_units / dt
        # algorithmic HBM bytes per locus: float64 read tensor in, packed trace + llk out
        bytes_in = args.reads * args.snvs * 2 * 8
        bytes_out = args.chains * args.mcmc_steps * (args.ploidy * 8 + 8)
        bytes_per_launch = U * (bytes_in + bytes_out)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "loci/sec (whole node) for 1000-step MCMC, tetraploid 8-SNV loci, 1/2/4/8 GPUs",
            "value": value,
            "unit": "loci/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d synthetic tetraploid loci per GPU, %d SNVs, %d reads, %d MCMC steps x %d chains, burn %d, "
                            "posterior summary (BASELINE.json configs[1])" % (U, args.snvs, args.reads, args.mcmc_steps, args.chains, args.burn),
                "loci_per_gpu": U, "ploidy": args.ploidy, "snvs": args.snvs, "reads": args.reads,
                "mcmc_steps": args.mcmc_steps, "chains": args.chains, "burn": args.burn,
                "llk_cache": not args.no_cache, "parallelism": "loci sharded contiguously, %d rank(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": kern_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args),
                "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": kern_ms, "sampler_span_ms": span_ms,
                "note": "the sampler is bound by the latency of its serial sub-steps, not by HBM (DESIGN.md 4): the HBM "
                        "fraction is reported because the contract asks for it",
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline leg runs at N = 1 only
            cores, quota = usable_cores()
            out["cpu_baseline"] = cpu_baseline(args, cores, quota)
            if args.ploidy * args.snvs <= 64:
                conc, dgpm = call_concordance(args, batch)
                out["cpu_baseline"]["mode_call_concordance"] = conc
                out["cpu_baseline"]["max_abs_delta_gpm"] = dgpm
        print(json.dumps(out))
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
