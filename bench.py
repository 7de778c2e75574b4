#!/usr/bin/env python3
"""bench.py -- loci/sec of the MCMC haplotype assembler hot path on MI355X.

A "step" is one pass of the hot path (DenovoMCMC.fit for every unit of the batch, burn-in, posterior
summary) over one batch of synthetic loci already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1]: 10k synthetic tetraploid loci, 8 SNVs, 200 reads, 1000 MCMC steps (2 chains, burn 500).
With --gpus N every rank runs the same number of loci (weak scaling; --total-loci T shards T loci over the ranks
instead: strong scaling, BASELINE.json configs[2] with T = 100000).  Loci are independent, so the compute needs no
collective; the one exchange of the design -- the gather of the fixed-size posterior records of every rank (RCCL
all_gather over xGMI) -- is inside the timed region and reported as gather_ms.

Besides the headline line the JSON carries `value_incl_h2d` (the same pass fed from host memory with the compact int8 /
int16 input and the posterior records copied back: what a caller gets end to end) and, under `extra`, the other
single-GPU configurations of BASELINE.json: config4 (call-exact, hexaploid, 16 haplotypes, 500 reads) and config5
(octoploid, 20 SNVs, 1000 reads, 4 chains x 2000 steps), each with its kernel time, a roofline figure that means
something for it, and the oracle's CPU rate.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The timed region keeps several passes in flight on separate HIP streams.  The HIP runtime multiplexes a process's
# streams over GPU_MAX_HW_QUEUES hardware queues (default 4: with the null stream, two of four side streams then share
# a queue and run one after the other -- measured 868 k loci/s against 942 k with a queue each: 8 or 16); read at HIP start-up,
# so it is set here, before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_FLIGHTS = {}


def passes_in_flight(n):
    """One set of streams per process and count: every stream a process creates takes a share of its hardware queues
    (GPU_MAX_HW_QUEUES above), so the sections of this benchmark reuse the same ones."""
    from mchap_amd.device import PassesInFlight

    if n not in _FLIGHTS:
        _FLIGHTS[n] = PassesInFlight(n)
    return _FLIGHTS[n]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU).  Without WORLD_SIZE in the environment and N > 1 this process starts N fresh rank "
                         "processes (python -m torch.distributed.run) before anything touches the GPU and relays rank 0's line")
    ap.add_argument("--steps", type=int, default=50)   # 0.7 s of timed passes at the default workload
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loci", type=int, default=10000, help="loci per GPU")
    ap.add_argument("--mcmc-steps", type=int, default=1000)
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--burn", type=int, default=500)
    ap.add_argument("--reads", type=int, default=200)
    ap.add_argument("--snvs", type=int, default=8)
    ap.add_argument("--ploidy", type=int, default=4)
    ap.add_argument("--inflight", type=int, default=4,
                    help="passes in flight: each on its own HIP stream with its own device buffers, so that the thinly "
                         "occupied phases of one pass (prepare pass, hand-back rounds, tails) overlap the next pass")
    ap.add_argument("--no-cache", action="store_true", help="disable the per-chain llk cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units for the CPU baseline (0 = auto)")
    ap.add_argument("--total-loci", type=int, default=0, help="strong scaling: this many loci sharded over the ranks")
    ap.add_argument("--no-extras", action="store_true", help="skip value_incl_h2d and the config4 / config5 lines")
    ap.add_argument("--config5-loci", type=int, default=256)  # 1024 chains: one wave per SIMD
    ap.add_argument("--config4-units", type=int, default=256)
    ap.add_argument("--e2e-loci", type=int, default=1000)
    ap.add_argument("--call-units", type=int, default=4096, help="units of the `mchap call` sampler's extra (extra.call_mcmc)")
    ap.add_argument("--tempered", action="store_true",
                    help="also run extra.config5_tempered (one chain x 4 temperatures at configs[4]'s shape): minutes of GPU time, "
                         "so not part of the default line (profiles/ holds the measured run)")
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


def baseline_row(value, unit, cores, value_1_thread, sample, **more):
    """A cpu_baseline object: the rate on `cores` threads with the one-thread rate beside it, how the one scales to the other
    (`scaling` = value / (cores x value_1_thread); `scales` false below 0.6: then it is not a `cores`-thread figure), and the
    host's hardware threads beside the threads the box's cgroup grants."""
    scaling = value / (cores * value_1_thread) if value_1_thread and cores else None
    row = {"value": value, "unit": unit, "cores": cores, "host_threads": os.cpu_count(), "kind": "port", "value_1_thread": value_1_thread,
           "scaling": None if scaling is None else round(scaling, 3), "scales": bool(scaling is None or scaling >= 0.6), "sample": sample}
    row.update(more)
    return row


def cpu_rates(unit_fn, n_one, n_all, unit="units/s", what="", cores=None, cost=None):
    """The oracle (a C port of the reference's algorithm: kind "port") timed on ONE thread and on all usable threads, one unit per
    thread: unit_fn(i) runs unit i through oracle/mchap_oracle.c (ctypes releases the interpreter lock, so a thread pool of
    plain calls is real parallelism).  Every `extra` reports both rates, so that speed-ups are quoted against the same thing.
    cost: estimated cost of every unit (any scale) -- the units are then handed out longest first, so that one expensive unit
    started last does not set the multi-thread time (docs/example's units differ 100-fold).  `scaling` = the multi-thread rate
    over cores x the one-thread rate; below 0.6 the line says so (`scales`: false) -- such a baseline is not a 16-thread figure."""
    from concurrent.futures import ThreadPoolExecutor

    if cores is None:
        cores, _ = usable_cores()
    order1 = list(range(n_one))
    t = time.perf_counter()
    for i in order1:
        unit_fn(i)
    d1 = time.perf_counter() - t
    n_all = max(n_all, cores)
    order = list(range(n_all))
    if isinstance(cost, str) and cost == "measure":  # an untimed pass in the threads tells what every unit costs
        took = [0.0] * n_all

        def timed(i):
            t_ = time.perf_counter()
            unit_fn(i)
            took[i] = time.perf_counter() - t_

        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(timed, order))
        cost = took
    if cost is not None:
        order.sort(key=lambda i: -float(cost[i]))
    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(unit_fn, order))
    dn = time.perf_counter() - t
    row = baseline_row(n_all / dn, unit, cores, n_one / d1,
                       "%s: %d units on one thread (%.1f s), %d units on %d threads, one unit per thread%s (%.1f s)" % (
                           what, n_one, d1, n_all, cores, ", longest first" if cost is not None else "", dn))
    if cost is not None:
        # the two legs run different units: what the threads achieve is compared per unit of cost -- the time the units took
        # (or are estimated to take) over threads x wall time
        w1, wn = sum(float(cost[i]) for i in order1), sum(float(cost[i]) for i in order)
        if w1 > 0 and d1 > 0:
            sc = (wn / dn) / (cores * (w1 / d1))
            row["scaling"], row["scales"] = round(sc, 3), bool(sc >= 0.6)
    return row


def reference_postprocessing(args, batch, n=16):
    """BASELINE.md section 4's separate line: what the reference does to every unit's trace IN PYTHON, whatever ran the sampler --
    GenotypeMultiTrace.__post_init__ sorts the haplotypes of every recorded step with one np.lexsort call per (chain, step)
    (assemble/classes.py:265-278 through encoding/integer/sequence.py:78-110), and posterior() counts distinct genotypes after
    the burn-in (classes.py:307-325).  Restated here with the same numpy calls (a port, not the reference's file) and timed on
    one host thread over the raw traces of the first n units of the timed batch.  The build's counterpart is fused into the
    sampler's trace write plus trace_posterior_kernel (inside `value`)."""
    from mchap_amd.assemble import unpack_trace

    n = min(n, args.loci)
    words, fixed, _, _ = batch.traces()
    units = [unpack_trace(words[u], fixed[u], 2) for u in range(n)]  # int8 [chains, steps, ploidy, snvs]
    t = time.perf_counter()
    for g in units:
        g = g.copy()
        for c in range(g.shape[0]):
            for i in range(g.shape[1]):
                a = g[c, i]
                g[c, i] = a[np.lexsort(np.flip(a, axis=-1).transpose((-1, -2)))]
    d_sort = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for g in units:
        m = np.ascontiguousarray(g[:, args.burn:]).reshape(-1, g.shape[2] * g.shape[3])
        _, counts = np.unique(m.view([("", m.dtype)] * m.shape[1]), return_counts=True)
        probs = counts / counts.sum()
        np.flip(np.argsort(probs))
    d_post = (time.perf_counter() - t) / n
    return {"kind": "port", "cores": 1, "units": n, "lexsort_ms_per_unit": d_sort * 1e3, "posterior_ms_per_unit": d_post * 1e3,
            "implied_ceiling_loci_per_s_per_core": 1.0 / (d_sort + d_post),
            "note": "the reference's Python-side trace post-processing per unit (one np.lexsort per chain and step, then the histogram of "
                    "distinct genotypes), restated with the same numpy calls and timed on one host thread: with it a reference run "
                    "cannot exceed this rate per core however fast its numba sampler is; it is NOT part of cpu_baseline.value "
                    "(oracle: sampler only)"}


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
    return baseline_row(n / dt, "loci/s", cores, n1 / dt1,
                        "%d loci of the same workload, oracle/mchap_oracle.c with llk cache threshold 100, OpenMP over loci "
                        "(%d threads = affinity capped by the cgroup CPU quota %s), %.2f s wall"
                        % (n, cores, "none" if quota is None else "%.1f" % quota, dt),
                        llk_evals_per_locus=st.llk_evals / n, llk_cache_hits_per_locus=st.llk_cache_hits / n)


def _profile_order(path):
    """profiles/ are named r<round><tag>_...: order by round, then by the tag as a spreadsheet column (a..z, aa, ab, ...), so that
    r03ab (end of round 3) ranks above r03h (mid round) -- a plain string sort put 'h' above 'ab'."""
    import re

    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, "")


def kernel_sources_sha16():
    """First 16 hex digits of the SHA-256 over the library's sources (mchap_amd/csrc/*.hpp, *.hip, *.inc, *.cpp, Makefile, in name
    order): what a committed counter summary must carry to be quoted for the kernels this run executes."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(f for pat in ("*.hpp", "*.hip", "*.inc", "*.cpp", "Makefile") for f in glob.glob(os.path.join(ROOT, "mchap_amd", "csrc", pat)))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_profile(args):
    """The newest committed PMC summary that matches this workload (profiles/*_pmc_traffic.json, produced by
    tools/profile_round.sh + tools/pmc_traffic.py in separate rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE converted with
    factors calibrated in the sampler's access width, MI355X_MICROARCH.md "HBM"; SQ counters).  These are STATIC figures of an
    earlier profiling run of the same command, not measurements of this run: the line names the file they come from."""
    import glob

    sha = kernel_sources_sha16()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=_profile_order, reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("loci") == args.loci and t.get("mcmc_steps") == args.mcmc_steps and t.get("chains") == args.chains:
                # (round 5) only a summary taken from THESE kernel sources is quoted: counters of other kernels say nothing here
                if t.get("kernel_sources_sha16") == sha:
                    return t, "profiles/" + os.path.basename(path)
                stale = stale or os.path.basename(path)
        except Exception:
            pass
    return None, (None if stale is None else "none of the current kernel sources (%s); newest of other sources: profiles/%s, not quoted" % (sha, stale))


# Issue cycles of one wavefront instruction on a SIMD (MI355X_MICROARCH.md "Wave scheduling" / "Per-instruction cycle constants":
# a wave64 32-bit VALU op issues over 2 cycles once the SIMD has a second wave to interleave; float64 runs at half that rate
# (78.6 against 157.3 TFLOP/s) = 4 cycles, 64-bit integer ops are priced the same, transcendentals 8)
SIMD_CYCLES_PER_S = 1024 * 2.4e9  # 256 CUs x 4 SIMDs x 2.4 GHz
VALU_CYCLES = {"f64": 4.0, "int64": 4.0, "trans": 8.0, "other": 2.0}


def valu_issue_cycles(sq):
    """Issue cycles of a launch's wavefront VALU instructions, weighted per class from the SQ counters (None when the per-class
    counters are not in the summary)."""
    if not sq or "SQ_INSTS_VALU" not in sq:
        return None
    tot = float(sq["SQ_INSTS_VALU"])
    f64 = sum(float(sq.get(k, 0.0)) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
    i64 = float(sq.get("SQ_INSTS_VALU_INT64", 0.0))
    trans = sum(float(sq.get(k, 0.0)) for k in ("SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_TRANS_F16"))
    have = all(k in sq for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
    other = max(0.0, tot - f64 - i64 - trans)
    cyc = f64 * VALU_CYCLES["f64"] + i64 * VALU_CYCLES["int64"] + trans * VALU_CYCLES["trans"] + other * VALU_CYCLES["other"]
    return {"cycles": cyc, "f64": f64, "int64": i64, "trans": trans, "other": other, "per_class_complete": have}


def incl_h2d(args, model):
    """The same pass as the caller sees it from host memory: compact input (int8 allele calls + int16 base qualities,
    4.8 KB per locus at the default shape) uploaded, prepare + sampler + posterior summary, the posterior records
    (<= 4 KB per locus) copied back.  Returns loci/s including both PCIe legs (pageable host memory, as numpy gives)."""
    import torch
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    U = args.loci
    _, calls, _ = synth_units(U, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=0)
    rng = np.random.default_rng(5)
    quals = rng.integers(20, 41, size=calls.shape).astype(np.int16)

    def start():
        b = DenovoDeviceBatch(model, None, calls=calls, quals=quals)
        b.run()
        b.posterior(args.burn)
        return b

    def once():
        return start().posterior_host()

    once()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 3
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    out = {"value": U / dt, "unit": "loci/s", "ms_per_pass": dt * 1e3, "h2d_bytes_per_locus": int(calls[0].size * 3),
           "d2h_bytes_per_locus": int(32 * args.ploidy * 8 + 32 * 4 + 40),
           "note": "device buffers allocated, compact input uploaded, posterior records downloaded inside the timed region"}
    nfl = max(1, args.inflight)
    if nfl > 1:
        # the same with several passes in flight: pass i is uploaded and enqueued on stream i % nfl, its records are
        # downloaded when its stream comes round again -- the PCIe legs of one pass overlap the kernels of the others
        from mchap_amd.device import pinned

        calls, quals = pinned(calls), pinned(quals)  # (a loader would read into page-locked buffers: uploads are asynchronous)
        flight = passes_in_flight(nfl)
        slots = [None] * nfl

        def turn(i, n_pass):
            k = i % nfl
            with torch.cuda.stream(flight.streams[k]):
                if slots[k] is not None:
                    slots[k].posterior_host()
                    slots[k] = None
                if i < n_pass:
                    slots[k] = start()

        for i in range(2 * nfl):
            turn(i, nfl)
        torch.cuda.synchronize()
        n = 6 * nfl
        t = time.perf_counter()
        for i in range(n + nfl):
            turn(i, n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        out = dict(out, value_one_in_flight=out["value"], ms_per_pass_one_in_flight=out["ms_per_pass"], value=U / dt, ms_per_pass=dt * 1e3,
                   passes_in_flight=nfl)
        out["note"] += "; %d passes in flight on separate streams (one host thread, input in page-locked host memory)" % nfl
    return out


def _lib_exact_ws(U, H, K):
    from mchap_amd import _lib

    return _lib.lib().mchap_exact_workspace_bytes(U, H, K)


def bench_config4(args):
    """BASELINE.json configs[3]: call-exact, hexaploid, 16 known haplotypes over 10 SNVs, 500 reads (G = 54 264 genotypes),
    Dirichlet-multinomial prior; inputs resident in HBM; streaming form (mode, support, frequencies) and array form (GL + GP)."""
    import torch
    from mchap_amd.device import ExactDeviceBatch
    from mchap_amd.synth import synth_units

    U, K, H, M, R = args.config4_units, 6, 16, 10, 500
    rng = np.random.default_rng(4)
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10), first_unit=400)
    haps = np.unique(rng.integers(0, 2, size=(64, M)).astype(np.int8), axis=0)[:H]
    prior = (0.1, rng.dirichlet(np.ones(H)))
    batch = ExactDeviceBatch(reads, K, haps, None, prior)
    res = {}
    for name, kw in (("streaming", dict(streaming=True, arrays=False)), ("arrays", dict(streaming=False, arrays=True))):
        batch.run(**kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 3
        for _ in range(n):
            batch.run(**kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n
    G = batch.G
    # (genotype, read) terms: formed once when the workspace keeps llk + log prior of every genotype for the second pass
    # (mchap_exact_workspace_bytes_cached, ExactDeviceBatch's default), twice otherwise
    cached = batch.ws_bytes > int(_lib_exact_ws(U, H, K))
    terms = (1.0 if cached else 2.0) * U * G * R
    # Instructions per (genotype, read) term from the SQ counter passes of THIS workload on the current kernel sources
    # (profiles/*_config4_sq_counters.json, tools/profile_round.sh: exact_pass1_kernel of one pass over 256 units; a summary taken
    # from other sources is not quoted -- sq_counters_of).  float64 flop per term = FMA x 2 + ADD + MUL + reciprocals;
    # -ffp-contract=off keeps the K multiply-adds of a read's mean apart, as the reference's compiled loop has them.
    ms = res["streaming"]
    c4, c4_src, c4_names = sq_counters_of("*_config4_sq_counters.json", ("exact_pass1_kernel",))
    roof = {"bound": "valu_fp64", "peak": 78.6, "unit": "TFLOP/s", "terms_per_s": terms / (ms * 1e-3),
            "second_pass": "from the joint log-probabilities kept in the workspace" if cached else "recomputed",
            "note": "log-throughput bound: one float64 log per group of four (genotype, read) terms where the reads carry no counts "
                    "(round 5: read_log_product), else one per term; 80 KB in, < 1 KB out per unit"}
    if c4 is not None and c4.get("SQ_INSTS_VALU"):
        per_launch_terms = float(256) * G * R / 64.0  # wavefront-terms of one launch of the profiled workload (256 units)
        launches = max(1.0, round(float(c4["SQ_WAVES"]) / max(1.0, 256.0 * ((G + 4095) // 4096) * 4)))  # (4 waves per workgroup of 256 threads)
        wt = per_launch_terms * launches
        vpt = float(c4["SQ_INSTS_VALU"]) / wt
        fma, add, mul = (float(c4.get(k, 0.0)) / wt for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
        trans, i64 = float(c4.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) / wt, float(c4.get("SQ_INSTS_VALU_INT64", 0.0)) / wt
        flop_per_term = 2.0 * fma + add + mul + trans
        cyc = (fma + add + mul) * VALU_CYCLES["f64"] + trans * VALU_CYCLES["trans"] + i64 * VALU_CYCLES["int64"] \
            + max(0.0, vpt - fma - add - mul - trans - i64) * VALU_CYCLES["other"]
        roof.update({"achieved": terms * flop_per_term / (ms * 1e-3) / 1e12, "frac": terms * flop_per_term / (ms * 1e-3) / 1e12 / 78.6,
                     "flop_per_term": flop_per_term, "valu_insts_per_wave_term": vpt,
                     "valu_issue_frac": terms / 64.0 * cyc / (ms * 1e-3) / SIMD_CYCLES_PER_S, "valu_issue_cycles_per_wave_term": cyc,
                     "valu_issue_note": "issue cycles weighted per instruction class (float64 and 64-bit integer 4, transcendental 8, other 2 "
                                        "cycles per wavefront instruction) over 1024 SIMDs x 2.4 GHz",
                     "counters": "%s (%s; SQ_INSTS_VALU* over %d launch(es): static figures of an earlier run on these sources)" % (c4_src, c4_names, int(launches))})
    else:
        roof.update({"achieved": None, "frac": None, "counters": "no SQ counter summary of the current kernel sources under profiles/ "
                                                                   "(tools/profile_round.sh writes one): instruction figures not quoted"})
    out = {
        "workload": "%d units: hexaploid, %d haplotypes x %d SNVs, %d reads, G = %d genotypes, prior (0.1, Dirichlet(1)); HBM resident" % (U, H, M, R, G),
        "value": U / (ms * 1e-3), "unit": "units/s", "kernel": "exact_pass1_kernel + exact_pass2_kernel", "kernel_ms": ms,
        "arrays_value": U / (res["arrays"] * 1e-3), "arrays_kernel_ms": res["arrays"],
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        from oracle import binding as orc

        mode = batch.mode_results()

        def one(u):
            a, ml, mp, sp, fq, oc = orc.posterior_mode(reads[u % U], K, haps, None, prior)
            assert a.tolist() == mode[0][u % U].tolist() and abs(mp - mode[2][u % U]) < 1e-9

        out["cpu_baseline"] = cpu_rates(one, 2, 0, "units/s", "oracle/mchap_oracle.c posterior_mode; mode and GPM equal the GPU's")
    return out


def sq_counters_of(pattern, kernel_prefix):
    """SQ counter sums of the newest profiles/<pattern> summary (tools/pmc_sq.py output: {kernel: {counter: sum}}) over the kernels
    whose name starts with kernel_prefix (a string or a tuple of strings: a sampler call is several kernels)."""
    import glob

    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=_profile_order, reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if (t.get("_meta") or {}).get("kernel_sources_sha16") != kernel_sources_sha16():
                continue  # (round 5) counters of other kernel sources are not quoted
            tot, names = {}, []
            for name, c in t.items():
                if name.startswith(prefixes) and "SQ_INSTS_VALU" in c:
                    names.append(name)
                    for k, v in c.items():
                        if k != "launches_seen":
                            tot[k] = tot.get(k, 0.0) + float(v)
            if names:
                return tot, "profiles/" + os.path.basename(path), " + ".join(sorted(names))
        except Exception:
            pass
    return None, None, None


def config5_roofline(U, substeps, kms):
    """VALU issue at configs[4]: wavefront VALU instructions of the sampler's launches from the committed SQ counter passes of this
    workload at 256 loci (static figures of an earlier profiling run, scaled per sub-step), priced per class in issue cycles."""
    c, src, kname = sq_counters_of("*_config5_sq_counters.json", ("denovo_spec_kernel<8", "denovo_fillw_kernel<8", "mchap::denovo_coast_kernel",
                                                                   "denovo_coast_kernel"))
    if c is None:
        return None
    ref_substeps = 256.0 * 4 * 2000 * (8 * 20 + 3)
    per = float(c["SQ_INSTS_VALU"]) / ref_substeps
    w = valu_issue_cycles(c)
    cyc = w["cycles"] / ref_substeps
    return {"bound": "valu_issue", "achieved": cyc * substeps / (kms * 1e-3) / 1e9, "peak": SIMD_CYCLES_PER_S / 1e9,
            "unit": "G SIMD issue cycles/s", "frac": cyc * substeps / (kms * 1e-3) / SIMD_CYCLES_PER_S,
            "sub_steps_per_s": substeps / (kms * 1e-3), "valu_insts_per_sub_step": per, "issue_cycles_per_sub_step": cyc,
            "per_class_complete": w["per_class_complete"], "counters": "%s (%s; static: an earlier rocprofv3 --pmc run)" % (src, kname),
            "note": "issue cycles weighted per class (float64 / 64-bit integer 4, transcendental 8, other 2; when the summary lacks the "
                    "float64 ADD / MUL counters they are priced at 2 and the fraction is a lower bound).  One pass alone holds one "
                    "wavefront per SIMD and waits on memory (SQ_WAIT_ANY); batches in flight fill the rest (value)"}


def bench_config5(args):
    """BASELINE.json configs[4] on one GPU: octoploid, 20 SNVs, 1000 reads, 4 chains x 2000 steps (the LDS-pressure shape)."""
    import torch
    from mchap_amd import DenovoMCMC, _lib
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    U, K, M, R, C_, S = args.config5_loci, 8, 20, 1000, 4, 2000
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(8, 20), first_unit=77)
    model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=S, chains=C_, random_seed=42)
    batch = DenovoDeviceBatch(model, reads)
    batch.time_sampler(True)
    t = time.perf_counter()
    batch.run()
    batch.posterior(S // 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    kms = batch.sampler_ms()
    batch.time_sampler(False)
    status = batch.d_status.cpu().numpy()
    n = K * M
    substeps = float(U) * C_ * S * (n + 3)  # mutation sub-steps + the three structural compound steps, per chain step
    out = {
        "workload": "%d loci: octoploid, %d SNVs, %d reads, %d chains x %d steps, burn %d; HBM resident; one pass" % (U, M, R, C_, S, S // 2),
        "value": U / dt, "unit": "loci/s", "kernel": batch.sampler_name, "kernel_ms": kms, "pass_ms": dt * 1e3,
        "ok": bool((status <= 1).all()),
        "roofline": config5_roofline(U, substeps, kms),
    }
    nfl = 2 * args.inflight if args.inflight > 1 else 1
    if nfl > 1:
        # the same pass with several batches in flight (256 loci are 1024 chains: half of the chip's wavefront slots,
        # and at ploidy 8 a pass spends most of its time behind a thinning front of chains: twice the headline's count)
        batches = [batch] + [DenovoDeviceBatch(model, reads) for _ in range(nfl - 1)]
        flight = passes_in_flight(nfl)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for b in batches:
            flight.submit(lambda b=b: (b.run(), b.posterior(S // 2)))
        flight.join()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t
        same = all(torch.equal(b.d_trace, batch.d_trace) for b in batches[1:])
        out["value_one_in_flight"] = out["value"]
        out["value"] = nfl * U / dtf
        out["passes_in_flight"] = nfl
        out["pass_ms_in_flight"] = dtf * 1e3 / nfl
        out["workload"] = out["workload"].replace("one pass", "%d batches in flight on separate streams (kernel_ms / pass_ms: one alone)" % nfl)
        out["ok"] = bool(out["ok"] and same)
        del batches
    if not args.no_cpu_baseline:
        from oracle import binding as orc
        from mchap_amd.assemble import break_table

        cfg = orc.make_cfg(K, S, C_, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX, break_table=break_table(M, 1.0, 3.0))
        cores, _ = usable_cores()
        n_cpu = max(1, min(U, cores))
        t = time.perf_counter()
        orc.denovo_fit_batch(cfg, reads[:1], [2] * M, n_threads=1, keep_traces=False)
        d1 = time.perf_counter() - t
        t = time.perf_counter()
        orc.denovo_fit_batch(cfg, reads[:n_cpu], [2] * M, n_threads=cores, keep_traces=False)
        dc = time.perf_counter() - t
        out["cpu_baseline"] = baseline_row(n_cpu / dc, "loci/s", min(cores, n_cpu), 1.0 / d1,
                                           "oracle with llk cache: 1 locus on one thread (%.1f s), %d loci of the same workload one locus per "
                                           "thread (%.1f s wall)" % (d1, n_cpu, dc))
    return out


def call_mcmc_workload(U):
    """The extra.call_mcmc workload resident in HBM (also what tools/call_once.py profiles): -> (once, shapes, arrays)."""
    import ctypes as C
    import torch
    from mchap_amd import _lib
    from mchap_amd.synth import synth_units

    K, H, M, R, S, Cn = 4, 16, 8, 200, 2000, 2
    rng = np.random.default_rng(9)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=700)
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(8 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool)
        tr = np.unique(truth[u], axis=0)
        rest = [p_ for p_ in pool if not any(np.array_equal(p_, t) for t in tr)][: H - len(tr)]
        hs = np.concatenate([tr, np.array(rest, np.int8)])
        rng.shuffle(hs)
        haps[u] = hs
    F = np.full(U, 0.1)
    dev = torch.device("cuda", torch.cuda.current_device())
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    d_reads, d_haps, d_F = d(reads), d(haps), d(F)
    d_sid = d(np.arange(U, dtype=np.int64))
    d_g = torch.empty(U * Cn * S * K, dtype=torch.int64, device=dev)
    d_l = torch.empty(U * Cn * S, dtype=torch.float64, device=dev)
    d_st = torch.empty(U, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = int(L.mchap_call_mcmc_workspace_bytes_for(U, R, H, K, S, Cn))
    d_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    keep = (d_reads, d_haps, d_F, d_sid, d_l, d_ws)

    def once():
        _lib.check(L.mchap_call_mcmc_batch_device(U, p(d_reads), R, M, 2, None, p(d_haps), H, K, 1, p(d_F), None, None, p(d_sid), S, Cn, 0,
                                                  C.c_uint64(42), p(d_g), p(d_l), p(d_st), p(d_ws), C.c_int64(ws), C.c_void_p(stream)))

    return once, (K, H, M, R, S, Cn), dict(reads=reads, haps=haps, d_g=d_g, d_st=d_st, keep=keep)


def call_mcmc_roofline(U, S, Cn, K, ms):
    """VALU issue of call_mcmc_kernel from the committed SQ counter passes of this workload at 4096 units (static figures of an
    earlier rocprofv3 --pmc run, scaled per allele sub-step), priced per instruction class; None while no such profile exists."""
    c, src, kname = sq_counters_of("*_call_sq_counters.json", ("call_mcmc_kernel", "mchap::call_mcmc_kernel", "call_coast_kernel", "mchap::call_coast_kernel"))
    if c is None:
        return None
    ref = 4096.0 * 2 * 2000 * 4
    w = valu_issue_cycles(c)
    sub = float(U) * Cn * S * K
    cyc = w["cycles"] / ref
    out = {"bound": "valu_issue", "achieved": cyc * sub / (ms * 1e-3) / 1e9, "peak": SIMD_CYCLES_PER_S / 1e9, "unit": "G SIMD issue cycles/s",
           "frac": cyc * sub / (ms * 1e-3) / SIMD_CYCLES_PER_S, "valu_insts_per_allele_sub_step": float(c["SQ_INSTS_VALU"]) / ref,
           "issue_cycles_per_allele_sub_step": cyc, "per_class_complete": w["per_class_complete"],
           "counters": "%s (%s; static: an earlier rocprofv3 --pmc run)" % (src, kname),
           "note": "a chain starts on a wavefront of its own (call_mcmc_kernel: the likelihoods of its first contexts, lanes over reads) and, once a "
                   "step met only remembered contexts, runs a lane per chain (call_coast_kernel: a sub-step is a key, a look-up among eight, "
                   "a Philox draw and a search over 16 running sums); a step is a serial program of about a thousand instructions whatever the "
                   "lanes in use, so the launch lasts steps x that program and the chip's issue slots are mostly idle at 8 192 chains -- "
                   "throughput grows with the batch until the LDS (a chain's memo: 2.1 KB) is full at 60 chains a compute unit"}
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        out["wait_any_frac_of_wave_cycles"] = float(c["SQ_WAIT_ANY"]) / float(c["SQ_WAVE_CYCLES"])
    return out


def bench_call_mcmc(args):
    """`mchap call`: the Gibbs sampler over the genotypes of KNOWN haplotypes (calling/mcmc.py:15-453) for a batch of
    tetraploid units -- 16 known haplotypes over 8 SNVs, 200 reads, the program's defaults of 2000 steps (burn 1000) x 2 chains,
    Dirichlet-multinomial prior with inbreeding 0.1 -- resident in HBM, mchap_call_mcmc_batch_device on torch's stream; the
    oracle on the same Philox streams beside it (and the first units' traces compared with it)."""
    import torch

    U = args.call_units
    once, (K, H, M, R, S, Cn), A_ = call_mcmc_workload(U)
    reads, haps, d_g, d_st = A_["reads"], A_["haps"], A_["d_g"], A_["d_st"]
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 3
    for _ in range(n):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    assert (d_st.cpu().numpy() == 0).all()
    out = {"workload": "%d units: tetraploid, %d known haplotypes x %d SNVs, %d reads, Gibbs steps, %d steps x %d chains, prior (0.1, flat); HBM resident" % (U, H, M, R, S, Cn),
           "value": U / (ms * 1e-3), "unit": "units/s", "kernel": "call_mcmc_kernel + call_coast_kernel", "kernel_ms": ms,
           "allele_steps_per_s": U * Cn * S * K / (ms * 1e-3), "roofline": call_mcmc_roofline(U, S, Cn, K, ms)}
    if not args.no_cpu_baseline:
        from oracle import binding as orc

        g = d_g.cpu().numpy().reshape(U, Cn, S, K)

        def one(u):
            go, lo = orc.call_mcmc(reads[u], haps[u], K, steps=S, chains=Cn, step_type=0, prior=(0.1, None), rng_kind=orc.RNG_PHILOX, seed=42, stream_id=u)
            assert np.array_equal(go, g[u]), "call sampler trace of unit %d differs from the oracle's" % u

        out["cpu_baseline"] = cpu_rates(one, 8, 64, "units/s", "oracle/mchap_oracle.c call_mcmc; traces equal the GPU's")
    return out


def bench_config1(args):
    """BASELINE.json configs[0]: `mchap assemble` on docs/example (20 target loci x 22 tetraploid samples, 2-23 SNVs, 0-534
    read pairs) through the program itself -- application.assemble, the notebook's settings (2000 steps, burn 1000, 2 chains,
    seed 42) -- from the pileup fixture tests/golden/example_biparental.npz (the BAM files are not in the repository; the
    fixture holds what extract_read_variants yields for them).  End-to-end = read encoding + one ragged sampler launch +
    device-side summaries + record formatting."""
    from mchap_amd import application

    path = os.path.join(ROOT, "tests", "golden", "example_biparental.npz")
    samples, targets, variants, matrices, contigs = application.load_matrices(path)

    class NSeq:
        known = False  # (application.assemble: the variants' REF alleles are written at their positions)

        def __getitem__(self, sl):
            return "N" * (sl.stop - sl.start)

    def once(timings):
        source = application.MatrixSource(samples, matrices)
        return list(application.assemble(None, variants, {c: NSeq() for c, _ in contigs}, source, ploidy=4, steps=2000, burn=1000, chains=2,
                                         seed=42, targets=targets, timings=timings))

    once({})
    tm = {}
    t = time.perf_counter()
    lines = once(tm)
    dt = time.perf_counter() - t
    # The same pileups as a job of a realistic size: the 20 targets sixteen times over (7 040 units in one ragged launch).
    # The 440 units alone are 880 chains -- less than half of the chip's wavefront slots, so that launch lasts as long as its
    # slowest chain; at scale the rate is set by the average chain.
    rep = 16
    tm16 = {}
    source16 = application.MatrixSource(samples, matrices)
    t = time.perf_counter()
    lines16 = list(application.assemble(None, variants, {c: NSeq() for c, _ in contigs}, source16, ploidy=4, steps=2000, burn=1000, chains=2,
                                        seed=42, targets=list(targets) * rep, timings=tm16))
    dt16 = time.perf_counter() - t
    assert lines16[:len(lines)] == lines and lines16[-len(lines):] == lines
    out = {
        "workload": "docs/example bi-parental population: %d target loci x %d samples = %d (locus x sample) units, tetraploid, 2-23 SNVs, "
                    "0-534 read pairs (de-duplicated rows + counts), 2000 steps x 2 chains, burn 1000; python program end to end from "
                    "the pileup fixture" % (len(targets), len(samples), tm["units"]),
        "value": len(lines) / dt, "unit": "loci/s", "units_per_s": tm["units"] / dt, "wall_ms": dt * 1e3,
        "sampler_units_per_s": tm["units"] / max(tm["sampler_s"], 1e-9), "sampler_ms": tm["sampler_s"] * 1e3,
        "encode_ms": tm["encode_s"] * 1e3, "format_ms": tm["format_s"] * 1e3, "records": len(lines),
        "note": "sampler_ms: one ragged launch (440 units of 20 shapes), posterior summary and MCI on the device, results on the host",
        "at_scale": {"workload": "the same 20 targets x %d: %d units" % (rep, tm16["units"]), "value": len(lines16) / dt16, "unit": "loci/s",
                     "units_per_s": tm16["units"] / dt16, "wall_ms": dt16 * 1e3, "sampler_units_per_s": tm16["units"] / max(tm16["sampler_s"], 1e-9),
                     "sampler_ms": tm16["sampler_s"] * 1e3, "encode_ms": tm16["encode_s"] * 1e3, "format_ms": tm16["format_s"] * 1e3},
    }
    if not args.no_cpu_baseline:
        from oracle import binding as orc
        from mchap_amd import io
        from mchap_amd.assemble import break_table

        source = application.MatrixSource(samples, matrices)
        todo = []
        for contig, start, stop, name in targets:
            locus = io.DenovoLocus(contig, start, stop, name, variants, "N" * (stop - start), sequence_known=False)
            M = len(locus.positions)
            for s_ in samples:
                sr = source.reads(locus, s_)
                if M and len(sr["dists"]):
                    todo.append((M, sr, list(locus.n_alleles)))

        def one(i):
            M, sr, n_alleles = todo[i % len(todo)]
            cfg = orc.make_cfg(4, 2000, 2, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX, stream_id=0,
                               break_table=break_table(M, 1.0, 3.0))
            # (the locus's own allele counts: the oracle's trie cache has n_alleles branches per position)
            assert orc.denovo_fit(cfg, sr["dists"], n_alleles, sr["counts"])[2] == 0

        n4 = sum(1 for _ in range(4 * len(samples)))
        # one thread: the units of the first 4 targets; all threads: EVERY unit of the example (the same job the GPU ran)
        out["cpu_baseline"] = cpu_rates(one, min(n4, len(todo)), len(todo), "units/s", "oracle/mchap_oracle.c, sampler only", cost="measure")
    return out


def _sampler_rate(model, reads, counts=None, reps=3, burn=None):
    """loci/s of DenovoRaggedBatch passes (sampler + device summaries; inputs resident) over ragged units, one at a time."""
    import torch
    from mchap_amd.device import DenovoRaggedBatch

    units = [dict(reads=r, counts=None if counts is None else counts[i], n_alleles=list(model.n_alleles), ploidy=int(model.ploidy),
                  inbreeding=None, stream_id=i) for i, r in enumerate(reads)]
    b = DenovoRaggedBatch(model, units)
    burn = model.steps // 2 if burn is None else burn
    b.run(burn)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        b.run(burn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    status = b.d_status.cpu().numpy()
    return len(units) / dt, dt * 1e3, bool((status <= 1).all())


def bench_config2_dedup(args):
    """SURVEY 8d's secondary variant of configs[1]: base phred scores ignored (the programs' default) -> every read row is
    one of a few dozen distinct rows, de-duplicated with read_counts (application/baseclass.py:207)."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import dedup_unit, synth_units

    U = args.loci
    reads, _, _ = synth_units(U, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, dedup=True)
    pairs = [dedup_unit(r) for r in reads]
    rows = [len(p[0]) for p in pairs]
    model = DenovoMCMC(ploidy=args.ploidy, n_alleles=[2] * args.snvs, steps=args.mcmc_steps, chains=args.chains, random_seed=42)
    rate, ms, ok = _sampler_rate(model, [p[0] for p in pairs], [p[1] for p in pairs], burn=args.burn)
    out = {"workload": "%d loci of configs[1]'s shape with base qualities ignored: %d reads -> %d-%d distinct rows (median %d) with counts; "
                       "ragged batch, HBM resident, one pass at a time" % (U, args.reads, min(rows), max(rows), int(np.median(rows))),
           "value": rate, "unit": "loci/s", "pass_ms": ms, "ok": ok}
    if not args.no_cpu_baseline:
        from oracle import binding as orc
        from mchap_amd.assemble import break_table

        cfg = orc.make_cfg(args.ploidy, args.mcmc_steps, args.chains, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX,
                           break_table=break_table(args.snvs, 1.0, 3.0))
        def one(i):
            assert orc.denovo_fit(cfg, pairs[i % len(pairs)][0], [2] * args.snvs, pairs[i % len(pairs)][1])[2] == 0

        out["cpu_baseline"] = cpu_rates(one, 64, 1024, "loci/s", "oracle with llk cache")
    return out


def bench_moving(args):
    """Chains that never settle (shallow, low-quality pileups: 16 and 40 reads of base quality 3-20): the regime in which the
    phased sampler's coasting cannot help."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    out = {}
    U = args.loci
    for R in (16, 40):
        reads, _, _ = synth_units(U, ploidy=args.ploidy, n_pos=args.snvs, n_reads=R, qual=(3, 20))
        model = DenovoMCMC(ploidy=args.ploidy, n_alleles=[2] * args.snvs, steps=args.mcmc_steps, chains=args.chains, random_seed=42)
        rate, ms, ok = _sampler_rate(model, list(reads), reps=1, burn=args.burn)
        row = {"workload": "%d loci of configs[1]'s shape with %d reads of base quality 3-20: every chain keeps moving" % (U, R),
               "value": rate, "unit": "loci/s", "pass_ms": ms, "ok": ok}
        if not args.no_cpu_baseline:
            from oracle import binding as orc
            from mchap_amd.assemble import break_table

            cfg = orc.make_cfg(args.ploidy, args.mcmc_steps, args.chains, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX,
                               break_table=break_table(args.snvs, 1.0, 3.0))
            cores, _ = usable_cores()
            n_cpu = 8 * cores
            t = time.perf_counter()
            _, _, code, _ = orc.denovo_fit_batch(cfg, reads[:n_cpu], [2] * args.snvs, n_threads=cores, keep_traces=False)
            dc = time.perf_counter() - t
            assert code == 0
            n1 = 16
            t = time.perf_counter()
            orc.denovo_fit_batch(cfg, reads[:n1], [2] * args.snvs, n_threads=1, keep_traces=False)
            d1 = time.perf_counter() - t
            row["cpu_baseline"] = baseline_row(n_cpu / dc, "loci/s", cores, n1 / d1,
                                               "oracle with llk cache: %d loci on %d threads, %d loci on one thread" % (n_cpu, cores, n1))
        out["reads_%d" % R] = row
    return out


def bench_config5_tempered(args, steps=2000, loci=None):
    """configs[4]'s secondary variant: one chain under the temperature ladder (0.001, 0.01, 0.1, 1.0), the replicas side
    by side, one wavefront each (denovo_spec_kernel<.., TW>).  The default line runs it REDUCED -- `steps` = 200 of the 2000 MCMC
    steps (a chain's steps cost the same throughout: the hot replicas never settle), reported per chain step and scaled to the full
    run -- with the oracle on the same reduced workload; `--tempered` runs the full 2000 steps."""
    import torch
    from mchap_amd import DenovoMCMC
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    U, K, M, R, S = (loci or args.config5_loci), 8, 20, 1000, steps
    temps = (0.001, 0.01, 0.1, 1.0)
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(8, 20), first_unit=77)
    model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=S, chains=1, temperatures=temps, random_seed=42)
    batch = DenovoDeviceBatch(model, reads)
    batch.time_sampler(True)
    t = time.perf_counter()
    batch.run()
    batch.posterior(S // 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    status = batch.d_status.cpu().numpy()
    out = {"workload": "%d loci: octoploid, %d SNVs, %d reads, 1 chain x 4 temperatures (0.001, 0.01, 0.1, 1.0) x %d steps%s; HBM resident; one pass"
                       % (U, M, R, S, "" if S == 2000 else " (REDUCED from 2000)"),
           "value": U / dt, "unit": "loci/s (runs of %d steps)" % S, "chain_steps_per_s": U * S / dt,
           "loci_per_s_scaled_to_2000_steps": U * S / dt / 2000.0,
           "kernel": batch.sampler_name + " (one wavefront per replica)", "kernel_ms": batch.sampler_ms(), "pass_ms": dt * 1e3,
           "ok": bool((status <= 1).all())}
    if not args.no_cpu_baseline and S <= 400:
        from oracle import binding as orc
        from mchap_amd.assemble import break_table

        cfg = orc.make_cfg(K, S, 1, None, temps, llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX, break_table=break_table(M, 1.0, 3.0))
        cores, _ = usable_cores()
        t = time.perf_counter()
        orc.denovo_fit_batch(cfg, reads[:1], [2] * M, n_threads=1, keep_traces=False)
        d1 = time.perf_counter() - t
        n_cpu = min(U, cores)
        t = time.perf_counter()
        orc.denovo_fit_batch(cfg, reads[:n_cpu], [2] * M, n_threads=cores, keep_traces=False)
        dc = time.perf_counter() - t
        out["cpu_baseline"] = baseline_row(n_cpu / dc, out["unit"], min(cores, n_cpu), 1.0 / d1,
                                           "oracle with llk cache on the same reduced workload: 1 locus on one thread (%.1f s), %d loci one per "
                                           "thread (%.1f s wall)" % (d1, n_cpu, dc))
    return out


def bench_program_e2e(args):
    """End-to-end rate of the `assemble` program on alignment files: a synthetic job on disk -- one
    indexed BAM of `--e2e-loci` target loci x 100 reads (tetraploid, 8 SNVs per locus), BED, VCF, FASTA -- through
    mchap_amd.application.assemble with the reference's default settings (2000 steps, burn 1000, 2 chains, qualities
    ignored -> de-duplicated rows with counts).  Reported: loci/s of the whole program, and where the time goes (BAM inflate
    + parse into columns, read extraction + encoding, sampler launch to results on the host, record formatting)."""
    import shutil
    import tempfile

    from mchap_amd import application, io, synth

    d = tempfile.mkdtemp(prefix="mchap_e2e_")
    try:
        t = time.perf_counter()
        job = synth.synth_assembly_inputs(d, n_loci=args.e2e_loci, n_samples=1, reads_per_locus=100)
        t_make = time.perf_counter() - t
        nbytes = os.path.getsize(job["bams"][0])

        def once(tm):
            t0 = time.perf_counter()
            source = application.ReadSource(io.sample_bam_table(job["bams"]), workers=4)
            tm["read_s"] = time.perf_counter() - t0
            ref = io.Reference(job["fasta"])
            return list(application.assemble(job["bed"], job["vcf"], ref, source, ploidy=4, steps=2000, burn=1000, chains=2, seed=42, timings=tm))

        once({})
        tm = {}
        t = time.perf_counter()
        lines = once(tm)
        dt = time.perf_counter() - t
        return {
            "workload": "%d synthetic target loci x 1 sample, 100 reads per locus in one indexed BAM (%.1f MB), tetraploid, 8 SNVs, 2000 steps x 2 "
                        "chains, burn 1000: files -> VCF records through application.assemble" % (args.e2e_loci, nbytes / 1e6),
            "value": len(lines) / dt, "unit": "loci/s", "wall_ms": dt * 1e3, "bam_read_ms": tm["read_s"] * 1e3, "encode_ms": tm["encode_s"] * 1e3,
            "sampler_ms": tm["sampler_s"] * 1e3, "format_ms": tm["format_s"] * 1e3, "records": len(lines),
            "bam_parse_ms": tm.get("bam_parse_s", 0.0) * 1e3, "device_wait_ms": tm.get("device_wait_s", 0.0) * 1e3,
            "note": "host work of the block as array operations (mchap_amd/blockpath.py): BGZF blocks inflated by 4 threads, records walked once by "
                    "the library (mchap_bam_columns), extraction / calls / de-duplication / descriptors / posterior haplotypes over all loci at once; "
                    "encode_ms includes bam_parse_ms, sampler_ms = launch + device_wait_ms (the launch lasts as long as its slowest chain) + "
                    "summaries; synthetic inputs written in %.1f s (not timed)" % t_make,
        }
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it (the reference deals its loci to a pool of worker processes the same
    way: application/baseclass.py:360-388): start N fresh rank processes -- this parent has not imported torch nor made any HIP
    call, so nothing that has initialised the GPU is ever re-executed --, relay what they print (rank 0 prints the one JSON line)
    and return their exit status."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus is not None and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % n_dev)
    dist = None
    backend = None
    # (MCHAP_BENCH_DIST=1: join a process group even as a single rank -- the RCCL calls of the N > 1 path on a 1-GPU box)
    if world > 1 or os.environ.get("MCHAP_BENCH_DIST") == "1":
        import torch.distributed as dist

        # RCCL over xGMI when every rank has a GPU of its own.  With fewer GPUs than ranks (the 1-GPU test boxes) the ranks share
        # devices and RCCL refuses two ranks on one device: gloo then carries the records (MCHAP_BENCH_BACKEND overrides)
        backend = os.environ.get("MCHAP_BENCH_BACKEND", "nccl" if n_dev >= world else "gloo")
        dist.init_process_group(backend=backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from mchap_amd import DenovoMCMC
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    from mchap_amd.shard import gather_records, shard_range

    if args.total_loci:
        first, stop = shard_range(args.total_loci, rank, world)  # strong scaling: contiguous, balanced shards
        U = stop - first
        n_total = args.total_loci
    else:
        U = args.loci
        first = rank * U  # rank r owns loci [r*U, (r+1)*U): contiguous shard, RNG keyed by global locus id
        n_total = U * world
    reads, _, _ = synth_units(U, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=first)
    model = DenovoMCMC(ploidy=args.ploidy, n_alleles=[2] * args.snvs, steps=args.mcmc_steps, chains=args.chains,
                       random_seed=42, llk_cache_threshold=-1 if args.no_cache else 100)
    nfl = max(1, args.inflight)
    batches = [DenovoDeviceBatch(model, reads, first_stream=first) for _ in range(nfl)]
    streams = passes_in_flight(nfl).streams if nfl > 1 else [torch.cuda.current_stream()]
    batch = batches[0]
    del reads

    kernel_ms = []
    gather_ev = []
    K = args.ploidy
    gathered = [None]

    def records(b):
        """The unit's posterior record as one int64 row: mode genotype words, (SPM, GPM) bit patterns, distinct states,
        count of the mode genotype -- what the application formats a VCF sample column from."""
        P = b.post
        return torch.cat([P["mode_words"].view(U, K), P["stats"].view(U, 2).view(torch.int64), P["n"].to(torch.int64).view(U, 1),
                          P["mode_count"].to(torch.int64).view(U, 1)], dim=1)

    def one_pass(i, events=None, isolated=False):
        """Pass i: prepare pass + sampler + posterior summary (+ the records' all-gather) enqueued on stream i % nfl."""
        b = batches[0] if isolated else batches[i % nfl]
        with torch.cuda.stream(streams[0] if isolated else streams[i % nfl]):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            b.run()
            if events is not None:
                e1.record()
                events.append((e0, e1))
            b.posterior(args.burn)
            if dist is not None:
                # the design's only exchange: every rank's records to every rank (padded all_gather; RCCL on GPUs)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                rec = records(b)
                if dist.get_backend() != "nccl":
                    rec = rec.cpu()
                gathered[0] = gather_records(rec, n_total, dist) if args.total_loci else _gather_equal(rec, dist)
                g1.record()
                if events is not None:
                    gather_ev.append((g0, g1))
        if isolated:
            kernel_ms.append(b.sampler_ms())  # waits for that launch only

    def _gather_equal(rec, dist_):
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist_.all_gather(parts, rec)
        return torch.cat(parts, dim=0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, nfl if args.warmup else 0)):  # (every batch's buffers touched once)
        one_pass(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_pass(i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # The kernel's own launch duration (roofline): one pass in flight, HIP events on the launch stream right around
    # the sampler's launches -- with several passes in flight those spans overlap and say nothing about the kernel
    batches[0].time_sampler(True)  # (per-call events, owned by this batch: nothing process-wide)
    events = []
    n_iso = max(1, min(args.steps, 10))
    one_pass(0, isolated=True)
    kernel_ms.clear()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(n_iso):
        one_pass(i, events, isolated=True)
    torch.cuda.synchronize()
    dt_iso = time.perf_counter() - t1
    batches[0].time_sampler(False)
    span_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))  # prepare pass + sampler + memsets
    kern_ms = float(np.mean(kernel_ms))
    kern_name = batches[0].sampler_name

    for b in batches:
        status = b.d_status.cpu().numpy()
        if (status > 1).any() or (status < 0).any():
            raise SystemExit("sampler reported errors: %s" % np.unique(status))
    for b in batches[1:]:  # every batch in flight produced the same traces
        if not (torch.equal(b.d_trace, batch.d_trace) and torch.equal(b.post["mode_words"], batch.post["mode_words"])):
            raise SystemExit("passes in flight disagree")
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ev])) if gather_ev else None
    gather_checked = None
    if dist is not None and rank == 0:
        # the gathered records of ANOTHER rank's units must be what a single-rank run of those units gives
        other = world - 1
        o_first = shard_range(n_total, other, world)[0] if args.total_loci else other * U
        n_chk = 8
        rd, _, _ = synth_units(n_chk, ploidy=args.ploidy, n_pos=args.snvs, n_reads=args.reads, first_unit=o_first)
        chk = DenovoDeviceBatch(model, rd, first_stream=o_first)
        chk.run()
        chk.posterior(args.burn)
        torch.cuda.synchronize()
        mine = chk.post["mode_words"].view(n_chk, K).cpu()
        got = gathered[0][o_first: o_first + n_chk, :K].cpu()
        if not torch.equal(mine, got):
            raise SystemExit("gathered posterior records differ from a single-rank run of the same units")
        gather_checked = n_chk

    if rank == 0:
        total_units = n_total * args.steps
        value = total_units / dt
        # algorithmic HBM bytes per locus: float64 read tensor in, packed trace + llk out
        bytes_in = args.reads * args.snvs * 2 * 8
        bytes_out = args.chains * args.mcmc_steps * (args.ploidy * 8 + 8)
        bytes_per_launch = U * (bytes_in + bytes_out)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        # the bound that binds: VALU issue (wavefront VALU instructions of a sampler call, from the SQ counter passes, against
        # 1024 SIMDs x 2.4 GHz / 4 cycles per 64-lane instruction), reported beside the HBM fraction the contract asks for
        prof, prof_src = pmc_profile(args)
        sq = (prof or {}).get("sq_counters_per_launch")
        w = valu_issue_cycles(sq)
        valu_insts = float(sq["SQ_INSTS_VALU"]) if w else None
        valu_frac = (w["cycles"] / (kern_ms * 1e-3) / SIMD_CYCLES_PER_S) if w else None
        out = {
            "metric": "loci/sec (whole node) for 1000-step MCMC, tetraploid 8-SNV loci, 1/2/4/8 GPUs",
            "value": value,
            "unit": "loci/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_loci else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d synthetic tetraploid loci per GPU, %d SNVs, %d reads, %d MCMC steps x %d chains, burn %d, "
                            "posterior summary (BASELINE.json configs[1]); %d passes in flight on separate HIP streams, each with its own "
                            "device buffers" % (U, args.snvs, args.reads, args.mcmc_steps, args.chains, args.burn, nfl),
                "loci_per_gpu": U, "ploidy": args.ploidy, "snvs": args.snvs, "reads": args.reads,
                "mcmc_steps": args.mcmc_steps, "chains": args.chains, "burn": args.burn,
                "llk_cache": not args.no_cache, "passes_in_flight": nfl,
                "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                "parallelism": "loci sharded contiguously over %d rank(s) on %d device(s); posterior records all-gathered every pass%s"
                               % (world, min(world, n_dev), "" if backend is None else " (%s)" % ("RCCL" if backend == "nccl" else backend)),
                "world_size": world, "devices": min(world, n_dev), "backend": backend,
            },
            "roofline": {
                "bound": "hbm", "kernel": kern_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": (prof or {}).get("hbm_bytes_per_launch"),
                "traffic_source": prof_src if prof is None else "%s: rocprofv3 --pmc passes of an EARLIER run of this command on the same "
                                                                "kernel sources (%s; static figure, not measured in this run)" % (prof_src, kernel_sources_sha16()),
                "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": kern_ms, "sampler_span_ms": span_ms,
                "valu_issue_frac": valu_frac, "valu_insts_per_launch": valu_insts, "valu_issue_cycles_per_launch": w["cycles"] if w else None,
                "valu_classes": None if not w else {k: w[k] for k in ("f64", "int64", "trans", "other", "per_class_complete")},
                "valu_counters": None if not w else "%s (static, as traffic); issue cycles = float64 x 4 + 64-bit integer x 4 + "
                                                    "transcendental x 8 + other x 2 over 1024 SIMDs x 2.4 GHz" % prof_src,
                "note": "a sampler call is several launches (phased sampler: speculative kernel phases + coasting kernel, "
                        "DESIGN.md 4.2.2); kernel_ms is the span of HIP events around all of them, measured with ONE pass in "
                        "flight in %d passes right after the timed region (in the timed region passes overlap on separate "
                        "streams and a launch's span is not its cost).  It is bound by wave-wide " % n_iso +
                        "likelihood evaluations at two waves per SIMD (first phase) and Philox throughput (coasting), not by "
                        "HBM: the HBM fraction is reported because the contract asks for it",
            },
        }
        if dist is None:
            out["value_one_in_flight"] = U * n_iso / dt_iso
        if dist is not None:
            out["gather_ms"] = gather_ms
            out["gather_bytes_per_rank"] = int(U * (K + 4) * 8)
            out["gather_checked_units"] = gather_checked
        if world == 1 and not args.no_extras:
            out["value_incl_h2d"] = incl_h2d(args, model)
            out["extra"] = {"config1": bench_config1(args), "config2_dedup": bench_config2_dedup(args), "moving": bench_moving(args),
                            "config4": bench_config4(args), "config5": bench_config5(args)}
            out["extra"]["program_e2e"] = bench_program_e2e(args)
            out["extra"]["call_mcmc"] = bench_call_mcmc(args)
            out["extra"]["config5_tempered"] = bench_config5_tempered(args, steps=200)
            if args.tempered:
                out["extra"]["config5_tempered_full"] = bench_config5_tempered(args)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline leg runs at N = 1 only
            cores, quota = usable_cores()
            out["cpu_baseline"] = cpu_baseline(args, cores, quota)
            if args.ploidy * args.snvs <= 64:
                conc, dgpm = call_concordance(args, batch)
                out["cpu_baseline"]["mode_call_concordance"] = conc
                out["cpu_baseline"]["max_abs_delta_gpm"] = dgpm
            out["cpu_baseline"]["reference_python_postprocessing"] = reference_postprocessing(args, batch)
        print(json.dumps(out))
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
