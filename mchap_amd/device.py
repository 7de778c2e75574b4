"""Device-resident batches: inputs and outputs stay in HBM, kernels are enqueued on the caller's
torch stream through the `*_device` entry points of the C ABI.  torch is used for device memory,
streams and (in bench.py) torch.distributed only.
"""
import ctypes as C

import numpy as np

from . import _lib
from .assemble import DenovoMCMC, unpack_trace


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise _lib.MchapLibraryError("no MI355X visible to torch: the device path needs a GPU (no CPU fallback)")
    return torch


def pinned(a):
    """A copy of numpy array `a` in page-locked host memory (as a numpy array): uploads from it are asynchronous."""
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()


class PassesInFlight:
    """Round-robin issue of whole passes (one device batch each: prepare pass, sampler, posterior summary) over
    `n` HIP streams.  A pass is a few dozen launches and several of them occupy a fraction of the chip (the prepare
    pass, the hand-back rounds of the phased sampler, every launch's last wave round); with three or four passes in
    flight those phases overlap the first phase of the next pass: 13.8 -> 10.6 ms per pass at 10 000 tetraploid loci
    (DESIGN.md 6).  Each pass needs device buffers of its own (its DenovoDeviceBatch / DenovoRaggedBatch).  The HIP
    runtime shares GPU_MAX_HW_QUEUES (default 4) hardware queues among a process's streams: export
    GPU_MAX_HW_QUEUES=8 (or 16) before the process touches the GPU so that four streams and the null stream get one each.

        flight = PassesInFlight(4)
        for batch in batches:
            flight.submit(lambda b=batch: (b.run(), b.posterior(burn)))
        flight.join()            # torch's current stream then waits for every pass
    """

    def __init__(self, n=4):
        torch = _torch()
        self.torch = torch
        self.streams = [torch.cuda.Stream() for _ in range(max(1, int(n)))]
        self.issued = 0

    def submit(self, enqueue):
        """Call `enqueue()` with the next stream current; what it enqueues starts after the work already on torch's
        current stream (inputs uploaded there are seen)."""
        s = self.streams[self.issued % len(self.streams)]
        self.issued += 1
        s.wait_stream(self.torch.cuda.current_stream())
        with self.torch.cuda.stream(s):
            return enqueue()

    def join(self):
        cur = self.torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)


class _OwnBuffers:
    """A device batch owns its buffers (inputs, workspace, traces, summaries): two passes over it must not overlap, whatever
    streams they are issued on.  Every enqueueing method waits for the event the previous one left behind and leaves its
    own -- passes of ONE batch are serialised on the device, passes of different batches still run side by side
    (PassesInFlight)."""

    _last_use = None

    def _begin(self):
        s = self.torch.cuda.current_stream()
        if self._last_use is not None:
            s.wait_event(self._last_use)
        return s

    def _end(self):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream())
        self._last_use = ev


class DenovoDeviceBatch(_OwnBuffers):
    """A batch of uniformly shaped units resident on one GPU.

    reads : float64 [U, R, M, A] (numpy, copied once) ; read_counts : int64 [U, R] or None."""

    def __init__(self, model: DenovoMCMC, reads, read_counts=None, first_stream=0, device=None, calls=None, quals=None,
                 error_rate=0.0024):
        """reads: float64 [U, R, M, A]; or reads=None with calls int8 [U, R, M] (< 0 = gap), optional quals int16 of the
        same shape and the base error rate: the compact form the reference's encoders start from, turned into the
        probability tensor on the device (5x fewer bytes to upload, same traces).  The input arrays are copied on torch's
        current stream; when they live in page-locked host memory (mchap_amd.device.pinned) the copy does not block the
        host, and the caller keeps them unchanged until the stream has passed it."""
        torch = _torch()
        self.torch = torch
        self.model = model
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.d_calls = self.d_quals = self.d_qual_prob = None
        if reads is None:
            calls = np.ascontiguousarray(calls, dtype=np.int8)
            U, R, M = calls.shape
            A = int(np.max(model.n_alleles))
            reads = None
        else:
            reads = np.ascontiguousarray(reads, dtype=np.float64)
            U, R, M, A = reads.shape
        self.shape = (U, R, M, A)
        K, Cn, S = int(model.ploidy), int(model.chains), int(model.steps)
        self.K, self.Cn, self.S = K, Cn, S
        units = np.zeros(U, dtype=_lib.UNIT_DTYPE)
        idx = np.arange(U, dtype=np.int64)
        units["reads_off"] = idx * (R * M * A) if reads is not None else idx * (R * M)
        units["counts_off"] = idx * R if read_counts is not None else -1
        units["nalleles_off"] = 0
        units["initial_off"] = -1
        units["trace_off"] = idx * (Cn * S * K)
        units["llk_off"] = idx * (Cn * S)
        units["fixed_off"] = idx * M
        units["n_reads"], units["n_pos"], units["max_allele"], units["ploidy"] = R, M, A, K
        units["inbreeding"] = np.nan if model.inbreeding is None else float(model.inbreeding)
        units["stream_id"] = (first_stream + idx).astype(np.uint64)
        self.units_host = units
        dev = self.device
        self.d_units = torch.from_numpy(units.view(np.uint8).reshape(-1)).to(dev)
        if reads is not None:
            self.d_reads = torch.from_numpy(reads.reshape(-1)).to(dev, non_blocking=True)
        else:
            from .encoding import prob_of_qual

            self.d_reads = None
            self.d_calls = torch.from_numpy(calls.reshape(-1)).to(dev, non_blocking=True)
            if quals is not None:
                quals = np.ascontiguousarray(quals, dtype=np.int16)
                assert quals.shape == calls.shape
                self.d_quals = torch.from_numpy(quals.reshape(-1)).to(dev, non_blocking=True)
                table = prob_of_qual(np.arange(int(quals.max(initial=0)) + 1)) * (1.0 - error_rate)  # reference io/bam.py:280-288
            else:
                table = np.array([1.0 - error_rate])
            self.qual_prob_len = len(table)
            self.d_qual_prob = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(dev)
        self.d_counts = None if read_counts is None else torch.from_numpy(np.ascontiguousarray(read_counts, dtype=np.int64).reshape(-1)).to(dev)
        self.d_nalleles = torch.from_numpy(np.asarray(model.n_alleles, dtype=np.int8)).to(dev)
        self.d_trace = torch.empty(U * Cn * S * K, dtype=torch.int64, device=dev)
        self.d_llks = torch.empty(U * Cn * S, dtype=torch.float64, device=dev)
        self.d_fixed = torch.empty(U * M, dtype=torch.int8, device=dev)
        self.d_status = torch.empty(U, dtype=torch.int32, device=dev)
        self.cfg = model._cfg(M)
        L = _lib.lib()
        self.ws_bytes = int(L.mchap_denovo_workspace_bytes(C.byref(self.cfg), U, _lib.ptr(units)))
        if self.ws_bytes < 0:
            raise NotImplementedError("mchap_hip: unsupported unit shape")
        if int(L.mchap_denovo_trace_words_per_haplotype(C.byref(self.cfg), U, _lib.ptr(units))) != 1:
            raise NotImplementedError("DenovoDeviceBatch holds one trace word per haplotype: units wider than 62 SNVs / 64 bits go "
                                      "through DenovoMCMC.fit_batch or DenovoRaggedBatch")
        self.d_ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)
        self.post = None
        self.sampler_name = _lib.sampler_name(self.cfg, units)
        self.timer = None

    def time_sampler(self, on=True):
        """Bracket the sampler launches of every run() with HIP events on its stream; `sampler_ms()` reads the last span."""
        if on and self.timer is None:
            self.timer = _lib.SamplerTimer()
        self.cfg.timer = self.timer.handle if (on and self.timer is not None) else None

    def sampler_ms(self):
        return -1.0 if self.timer is None else self.timer.ms()

    def _p(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def run(self):
        """Enqueue the sampler on torch's current stream (no synchronisation)."""
        L = _lib.lib()
        stream = self._begin().cuda_stream
        self.cfg.cache_epoch = _lib.next_cache_epoch()  # (the call's own: one sequence for every library copy of the process)
        if self.d_reads is None:
            rc = L.mchap_denovo_fit_batch_calls_device(
                C.byref(self.cfg), self.shape[0], self._p(self.d_units), _lib.ptr(self.units_host), self._p(self.d_calls),
                self._p(self.d_quals), self._p(self.d_qual_prob), int(self.qual_prob_len), self._p(self.d_counts),
                self._p(self.d_nalleles), None, self._p(self.d_trace), self._p(self.d_llks), self._p(self.d_fixed),
                self._p(self.d_status), self._p(self.d_ws), C.c_int64(self.ws_bytes), C.c_void_p(stream))
            _lib.check(rc)
            self._end()
            return
        rc = L.mchap_denovo_fit_batch_device(
            C.byref(self.cfg), self.shape[0], self._p(self.d_units), _lib.ptr(self.units_host), self._p(self.d_reads),
            self._p(self.d_counts), self._p(self.d_nalleles), None, self._p(self.d_trace), self._p(self.d_llks),
            self._p(self.d_fixed), self._p(self.d_status), self._p(self.d_ws), C.c_int64(self.ws_bytes), C.c_void_p(stream))
        _lib.check(rc)
        self._end()

    def posterior(self, burn, max_states=32):
        """Enqueue the posterior summary of the traces written by run()."""
        torch = self.torch
        U = self.shape[0]
        if self.post is None or self.post["max_states"] != max_states:
            dev = self.device
            self.post = dict(
                max_states=max_states,
                words=torch.empty(U * max_states * self.K, dtype=torch.int64, device=dev),
                counts=torch.empty(U * max_states, dtype=torch.int32, device=dev),
                n=torch.empty(U, dtype=torch.int32, device=dev),
                stats=torch.empty(U * 2, dtype=torch.float64, device=dev),
                mode=torch.empty(U, dtype=torch.int32, device=dev),
                mode_words=torch.empty(U * self.K, dtype=torch.int64, device=dev),
                mode_count=torch.empty(U, dtype=torch.int32, device=dev),
            )
        P = self.post
        stream = self._begin().cuda_stream
        rc = _lib.lib().mchap_trace_posterior_batch_device(
            U, self._p(self.d_units), self.S, self.Cn, int(burn), self._p(self.d_trace), int(max_states), self.K,
            self._p(P["words"]), self._p(P["counts"]), self._p(P["n"]), self._p(P["stats"]), self._p(P["mode"]),
            self._p(P["mode_words"]), self._p(P["mode_count"]), C.c_void_p(stream))
        _lib.check(rc)
        self._end()

    def incongruence(self, burn, threshold=0.6):
        """Enqueue the replicate-incongruence code (MCI) of every unit's chains; returns the int32 device tensor
        (0 none, 1 incongruence, 2 putative CNV; reference GenotypeMultiTrace.replicate_incongruence)."""
        torch = self.torch
        U = self.shape[0]
        if getattr(self, "d_mci", None) is None:
            self.d_mci = torch.empty(U, dtype=torch.int32, device=self.device)
        stream = self._begin().cuda_stream
        rc = _lib.lib().mchap_trace_incongruence_batch_device(
            U, self._p(self.d_units), self.S, self.Cn, int(burn), self._p(self.d_trace), self.K, C.c_double(float(threshold)),
            self._p(self.d_mci), C.c_void_p(stream))
        _lib.check(rc)
        self._end()
        return self.d_mci

    # ---- results back on the host ----
    def traces(self):
        U, R, M, A = self.shape
        w = self.d_trace.cpu().numpy().view(np.uint64).reshape(U, self.Cn, self.S, self.K)
        fixed = self.d_fixed.cpu().numpy().reshape(U, M)
        llks = self.d_llks.cpu().numpy().reshape(U, self.Cn, self.S)
        status = self.d_status.cpu().numpy()
        return w, fixed, llks, status

    def genotypes(self, u, words=None, fixed=None):
        if words is None:
            words, fixed, _, _ = self.traces()
        return unpack_trace(words[u], fixed[u], self.shape[3])

    def posterior_host(self):
        P = self.post
        U = self.shape[0]
        ms = P["max_states"]
        n = P["n"].cpu().numpy()
        if (n < 0).any():
            # more distinct genotypes than the kernel keeps (counts were dropped): the summary is not the posterior
            raise _lib.MchapLibraryError(
                "posterior summary overflow in %d unit(s): more than 512 distinct genotypes after burn-in; "
                "summarise those units from traces() instead" % int((n < 0).sum()))
        return dict(
            words=P["words"].cpu().numpy().view(np.uint64).reshape(U, ms, self.K),
            counts=P["counts"].cpu().numpy().reshape(U, ms),
            n=n,
            stats=P["stats"].cpu().numpy().reshape(U, 2),
            mode=P["mode"].cpu().numpy(),
            mode_words=P["mode_words"].cpu().numpy().view(np.uint64).reshape(U, self.K),
            mode_count=P["mode_count"].cpu().numpy(),
        )


class ExactDeviceBatch:
    """A batch of exact-caller units that share a shape, resident on one GPU: inputs are uploaded once, every output of
    mchap_exact_call_batch_device stays in HBM until asked for.

    reads float64 [U, R, M, A]; haplotypes int8 [U, H, M] (or [H, M] for all units); read_counts int64 [U, R] or None;
    prior None | (inbreeding scalar or [U], frequencies None | [H] | [U, H])."""

    def __init__(self, reads, ploidy, haplotypes, read_counts=None, prior=None, device=None, cache_joint=True):
        torch = _torch()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        dev = self.device
        reads = np.ascontiguousarray(reads, dtype=np.float64)
        U, R, M, A = reads.shape
        haps = np.asarray(haplotypes, dtype=np.int8)
        if haps.ndim == 2:
            haps = np.broadcast_to(haps, (U,) + haps.shape)
        haps = np.array(haps)
        H = haps.shape[1]
        self.shape = (U, R, M, A, H, int(ploidy))
        from math import comb

        self.G = comb(H + int(ploidy) - 1, int(ploidy))
        self.d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
        self.d_haps = torch.from_numpy(haps.reshape(-1)).to(dev)
        self.d_counts = None if read_counts is None else torch.from_numpy(
            np.ascontiguousarray(read_counts, dtype=np.int64).reshape(-1)).to(dev)
        self.has_prior = 0 if prior is None else 1
        self.d_F = self.d_fr = None
        if prior is not None:
            F = np.array(np.broadcast_to(np.asarray(prior[0], dtype=np.float64), (U,)))
            self.d_F = torch.from_numpy(F).to(dev)
            if prior[1] is not None:
                fr = np.array(np.broadcast_to(np.asarray(prior[1], dtype=np.float64), (U, H)))
                self.d_fr = torch.from_numpy(fr.reshape(-1)).to(dev)
        self.ws_bytes = int(_lib.lib().mchap_exact_workspace_bytes(U, H, int(ploidy)))
        free, _ = torch.cuda.mem_get_info()
        if self.ws_bytes < 0 or self.ws_bytes > free:
            # beyond what the exact caller enumerates here (the reference has no limit but time): the programs write such a record
            # with FILTER=LIMIT (application._run_exact_groups)
            raise NotImplementedError("mchap_hip: ploidy %d over %d haplotypes: %s genotypes are beyond this build's exact caller "
                                      "(2^62 indices; the workspace must fit the device)" % (int(ploidy), H, self.G))
        if cache_joint:
            # room for llk + log prior of every genotype between the two passes of the streaming form (8 bytes each)
            # (taken only when it fits comfortably: else the plain workspace, whose second pass forms the values again)
            big = int(_lib.lib().mchap_exact_workspace_bytes_cached(U, H, int(ploidy)))
            free, _ = torch.cuda.mem_get_info()
            if big <= min(16 << 30, free // 2):
                self.ws_bytes = big
        self.d_ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)
        self.out = {}

    def _p(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _buf(self, name, n, dtype):
        t = self.out.get(name)
        if t is None:
            t = self.torch.empty(n, dtype=dtype, device=self.device)
            self.out[name] = t
        return t

    def run(self, streaming=True, arrays=False, llks64=False):
        """Enqueue the exact caller on torch's current stream: the streaming outputs (posterior_mode), and / or the
        array outputs (likelihoods, posteriors and their summaries)."""
        torch = self.torch
        U, R, M, A, H, K = self.shape
        o = _lib.ExactOut()
        if streaming:
            o.mode_alleles = self._buf("mode_alleles", U * K, torch.int64).data_ptr()
            for nm in ("mode_llk", "mode_prob", "support_prob"):
                setattr(o, nm, self._buf(nm, U, torch.float64).data_ptr())
            for nm in ("freqs", "occur"):
                setattr(o, nm, self._buf(nm, U * H, torch.float64).data_ptr())
        if arrays:
            o.llks = self._buf("llks", U * self.G, torch.float32).data_ptr()
            if llks64:
                o.llks64 = self._buf("llks64", U * self.G, torch.float64).data_ptr()
            o.posteriors = self._buf("posteriors", U * self.G, torch.float64).data_ptr()
            o.arr_mode_alleles = self._buf("arr_mode_alleles", U * K, torch.int64).data_ptr()
            for nm in ("arr_mode_prob", "arr_support_prob"):
                setattr(o, nm, self._buf(nm, U, torch.float64).data_ptr())
            for nm in ("arr_freqs", "arr_counts", "arr_occur"):
                setattr(o, nm, self._buf(nm, U * H, torch.float64).data_ptr())
        stream = torch.cuda.current_stream().cuda_stream
        rc = _lib.lib().mchap_exact_call_batch_device(
            U, self._p(self.d_reads), R, M, A, self._p(self.d_counts), self._p(self.d_haps), H, K, self.has_prior,
            self._p(self.d_F), self._p(self.d_fr), C.byref(o), self._p(self.d_ws), C.c_int64(self.ws_bytes), C.c_void_p(stream))
        _lib.check(rc)

    def _host(self, name, shape):
        return self.out[name].cpu().numpy().reshape(shape)

    def mode_results(self):
        """(alleles [U, K], llk [U], prob [U], support_prob [U], freqs [U, H], occur [U, H]) of the streaming form."""
        U, R, M, A, H, K = self.shape
        return (self._host("mode_alleles", (U, K)), self._host("mode_llk", (U,)), self._host("mode_prob", (U,)),
                self._host("support_prob", (U,)), self._host("freqs", (U, H)), self._host("occur", (U, H)))

    def array_results(self, with_arrays=True):
        U, R, M, A, H, K = self.shape
        res = dict(alleles=self._host("arr_mode_alleles", (U, K)), prob=self._host("arr_mode_prob", (U,)),
                   support_prob=self._host("arr_support_prob", (U,)), freqs=self._host("arr_freqs", (U, H)),
                   counts=self._host("arr_counts", (U, H)), occur=self._host("arr_occur", (U, H)))
        if with_arrays:
            res["llks"] = self._host("llks", (U, self.G))
            res["posteriors"] = self._host("posteriors", (U, self.G))
        return res


class DenovoRaggedBatch(_OwnBuffers):
    """Units of different shapes (loci with different numbers of SNVs and alleles, samples with different read depths,
    per-sample ploidy / inbreeding) in ONE sampler launch, with the posterior summary and the replicate incongruence
    taken on the device: what `mchap assemble` needs per (locus x sample) comes back as a few hundred bytes per unit.

    units: list of dicts with reads float64 [R, M, A] (R >= 1, M >= 1), counts int64 [R] or None, n_alleles int [M],
    ploidy int, inbreeding float or None, stream_id int.  Sampler settings from `model` (a DenovoMCMC; its ploidy /
    n_alleles / inbreeding fields are not used)."""

    n_runs = 0  # sampler launches issued by this class (the application tests assert O(1) per VCF)

    def __init__(self, model: DenovoMCMC, units, device=None):
        torch = _torch()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        dev = self.device
        self.model = model
        U = len(units)
        Cn, S = int(model.chains), int(model.steps)
        self.Cn, self.S = Cn, S
        desc = np.zeros(U, dtype=_lib.UNIT_DTYPE)
        r_parts, c_parts, a_parts = [], [], []
        r_off = c_off = a_off = t_off = l_off = f_off = 0
        self.Kmax = max(int(u["ploidy"]) for u in units)
        self.max_pos = max(u["reads"].shape[1] for u in units)
        for i, u in enumerate(units):
            rd = np.ascontiguousarray(u["reads"], dtype=np.float64)
            R, M, A = rd.shape
            assert R >= 1 and M >= 1 and len(u["n_alleles"]) == M
            K = int(u["ploidy"])
            D = desc[i]
            D["reads_off"] = r_off
            r_parts.append(rd.reshape(-1))
            r_off += rd.size
            if u.get("counts") is not None:
                cn = np.ascontiguousarray(u["counts"], dtype=np.int64)
                assert cn.shape == (R,)
                D["counts_off"] = c_off
                c_parts.append(cn)
                c_off += R
            else:
                D["counts_off"] = -1
            D["nalleles_off"] = a_off
            a_parts.append(np.asarray(u["n_alleles"], dtype=np.int8))
            a_off += M
            D["initial_off"] = -1
            D["trace_off"] = t_off
            t_off += Cn * S * K
            D["llk_off"] = l_off
            l_off += Cn * S
            D["fixed_off"] = f_off
            f_off += M
            D["n_reads"], D["n_pos"], D["max_allele"], D["ploidy"] = R, M, A, K
            F = u.get("inbreeding")
            D["inbreeding"] = np.nan if F is None else float(F)
            D["stream_id"] = int(u.get("stream_id", 0))
        self.units_host = desc
        self.n_units = U
        self.cfg = model._cfg(self.max_pos)
        # uint64 words per haplotype of the traces: 2 when the batch holds a unit of more than 62 SNVs / 64 bits per haplotype
        self.wph = int(_lib.lib().mchap_denovo_trace_words_per_haplotype(C.byref(self.cfg), U, _lib.ptr(desc)))
        if self.wph < 1:
            raise NotImplementedError("mchap_hip: unsupported unit shape (" + _lib.last_error() + ")")
        desc["trace_off"] *= self.wph
        t_off *= self.wph
        self.d_units = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
        self.d_reads = torch.from_numpy(np.concatenate(r_parts)).to(dev)
        self.d_counts = torch.from_numpy(np.concatenate(c_parts)).to(dev) if c_parts else None
        self.d_nalleles = torch.from_numpy(np.concatenate(a_parts)).to(dev)
        self.d_trace = torch.empty(t_off, dtype=torch.int64, device=dev)
        self.d_llks = torch.empty(l_off, dtype=torch.float64, device=dev)
        self.d_fixed = torch.empty(f_off, dtype=torch.int8, device=dev)
        self.d_status = torch.empty(U, dtype=torch.int32, device=dev)
        self.cfg = model._cfg(self.max_pos)
        self.ws_bytes = int(_lib.lib().mchap_denovo_workspace_bytes(C.byref(self.cfg), U, _lib.ptr(desc)))
        if self.ws_bytes < 0:
            raise NotImplementedError("mchap_hip: unsupported unit shape")
        self.d_ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)

    def _p(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _fit(self, stream):
        """The sampler launch on `stream` (a raw hipStream_t): from the float64 read tensors, or -- a batch built by from_calls --
        from int8 calls, the tensors formed on the device by the prepare pass (mchap_denovo_fit_batch_calls_device)."""
        L = _lib.lib()
        self.cfg.cache_epoch = _lib.next_cache_epoch()
        if self.d_reads is None:
            _lib.check(L.mchap_denovo_fit_batch_calls_device(
                C.byref(self.cfg), self.n_units, self._p(self.d_units), _lib.ptr(self.units_host), self._p(self.d_calls), None,
                self._p(self.d_qual_prob), 1, self._p(self.d_counts), self._p(self.d_nalleles), None, self._p(self.d_trace),
                self._p(self.d_llks), self._p(self.d_fixed), self._p(self.d_status), self._p(self.d_ws), C.c_int64(self.ws_bytes),
                C.c_void_p(stream)))
            return
        _lib.check(L.mchap_denovo_fit_batch_device(
            C.byref(self.cfg), self.n_units, self._p(self.d_units), _lib.ptr(self.units_host), self._p(self.d_reads), self._p(self.d_counts),
            self._p(self.d_nalleles), None, self._p(self.d_trace), self._p(self.d_llks), self._p(self.d_fixed),
            self._p(self.d_status), self._p(self.d_ws), C.c_int64(self.ws_bytes), C.c_void_p(stream)))

    @classmethod
    def from_calls(cls, model, calls, reads_off, n_reads, n_pos, max_allele, ploidy, counts, counts_off, n_alleles, nalleles_off,
                   inbreeding=None, error_rate=0.0024, device=None):
        """The same batch from the compact input (reference encoders: encoding/integer/transcode.py:16-77 with base qualities
        ignored, as the programs do by default), built with array operations only -- one row of every array per unit:
        calls int8 (all units' [n_reads, n_pos] matrices of allele calls, < 0 = gap, unit u at reads_off[u]), counts int64 (unit u's
        n_reads[u] counts at counts_off[u]; -1 = none), n_alleles int8 (unit u's n_pos[u] values at nalleles_off[u]; units may
        share them), inbreeding float [U] (NaN = none) or None."""
        self = cls.__new__(cls)
        torch = _torch()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        dev = self.device
        self.model = model
        n_reads, n_pos, ploidy = (np.asarray(x, dtype=np.int64) for x in (n_reads, n_pos, ploidy))
        U = len(n_reads)
        Cn, S = int(model.chains), int(model.steps)
        self.Cn, self.S = Cn, S
        assert U and (n_reads >= 1).all() and (n_pos >= 1).all()
        desc = np.zeros(U, dtype=_lib.UNIT_DTYPE)
        desc["reads_off"], desc["counts_off"], desc["nalleles_off"] = reads_off, counts_off, nalleles_off
        desc["initial_off"] = -1
        t = Cn * S * ploidy
        desc["trace_off"] = np.cumsum(t) - t
        desc["llk_off"] = np.arange(U, dtype=np.int64) * (Cn * S)
        desc["fixed_off"] = np.cumsum(n_pos) - n_pos
        desc["n_reads"], desc["n_pos"], desc["max_allele"], desc["ploidy"] = n_reads, n_pos, max_allele, ploidy
        desc["inbreeding"] = np.nan if inbreeding is None else inbreeding
        desc["stream_id"] = 0
        self.Kmax, self.max_pos = int(ploidy.max()), int(n_pos.max())
        self.units_host, self.n_units = desc, U
        self.cfg = model._cfg(self.max_pos)
        self.wph = int(_lib.lib().mchap_denovo_trace_words_per_haplotype(C.byref(self.cfg), U, _lib.ptr(desc)))
        if self.wph < 1:
            raise NotImplementedError("mchap_hip: unsupported unit shape (" + _lib.last_error() + ")")
        desc["trace_off"] *= self.wph
        self.d_units = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
        self.d_reads = None
        self.d_calls = torch.from_numpy(np.ascontiguousarray(calls, dtype=np.int8)).to(dev)
        self.d_qual_prob = torch.from_numpy(np.array([1.0 - error_rate], dtype=np.float64)).to(dev)
        self.d_counts = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).to(dev) if len(counts) else None
        self.d_nalleles = torch.from_numpy(np.ascontiguousarray(n_alleles, dtype=np.int8)).to(dev)
        self.d_trace = torch.empty(int(t.sum()) * self.wph, dtype=torch.int64, device=dev)
        self.d_llks = torch.empty(U * Cn * S, dtype=torch.float64, device=dev)
        self.d_fixed = torch.empty(int(n_pos.sum()), dtype=torch.int8, device=dev)
        self.d_status = torch.empty(U, dtype=torch.int32, device=dev)
        self.ws_bytes = int(_lib.lib().mchap_denovo_workspace_bytes(C.byref(self.cfg), U, _lib.ptr(desc)))
        if self.ws_bytes < 0:
            raise NotImplementedError("mchap_hip: unsupported unit shape")
        self.d_ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)
        return self

    def _host_summary(self, u, st, fixed, cache):
        """The summary of unit u by the host classes on its downloaded traces (assemble/classes.py:280-376 as the reference runs
        them): only for a unit whose chains visited more distinct genotypes than even the listed launch's LDS table holds --
        tens of thousands of steps.  cache: dict holding the batch's traces once most of its units come this way."""
        from .classes import GenotypeMultiTrace

        D = self.units_host[u]
        Ku, M, A, W = int(D["ploidy"]), int(D["n_pos"]), int(D["max_allele"]), self.wph
        fx = fixed[int(D["fixed_off"]): int(D["fixed_off"]) + M]

        def piece(name, dev, lo, n, view=None):
            if cache.get("whole"):
                if name not in cache:
                    a = dev.cpu().numpy()
                    cache[name] = a.view(view) if view is not None else a
                return cache[name][lo: lo + n]
            a = dev[lo: lo + n].cpu().numpy()
            return a.view(view) if view is not None else a

        w = piece("trace", self.d_trace, int(D["trace_off"]), self.Cn * self.S * Ku * W, np.uint64)
        w = w.reshape((self.Cn, self.S, Ku) + ((W,) if W > 1 else ()))
        g = unpack_trace(w, fx, A, W)
        lk = piece("llks", self.d_llks, int(D["llk_off"]), self.Cn * self.S).reshape(self.Cn, self.S)
        tr = GenotypeMultiTrace._from_sorted(g, lk).burn(self.burn)
        post = tr.posterior()
        sup = post.mode_genotype_support()
        mg, gp = sup.mode_genotype()
        return dict(genotypes=post.genotypes, probabilities=post.probabilities, spm=float(sup.probabilities.sum()), gpm=float(gp),
                    mode_genotype=mg, mci=int(tr.replicate_incongruence(self.incongruence_threshold)), status=st)

    def run(self, burn, max_states=512, incongruence_threshold=0.6):
        """Sampler, posterior summary and incongruence code, all enqueued on torch's current stream."""
        torch = self.torch
        dev = self.device
        U, K, W = self.n_units, self.Kmax, self.wph  # (W = 2: a batch of the general sampler, two words per haplotype -- round 5)
        stream = self._begin().cuda_stream
        L = _lib.lib()
        type(self).n_runs += 1
        self._fit(stream)
        self.max_states = max_states
        if W > 1:  # (wider states: the batch launch's table is what the LDS holds of them)
            max_states = min(int(max_states), int(L.mchap_trace_posterior_max_states_wph(K, W)))
            self.max_states = max_states
        self.p_words = torch.empty(U * max_states * K * W, dtype=torch.int64, device=dev)
        self.p_counts = torch.empty(U * max_states, dtype=torch.int32, device=dev)
        self.p_n = torch.empty(U, dtype=torch.int32, device=dev)
        self.p_stats = torch.empty(U * 2, dtype=torch.float64, device=dev)
        self.p_mode = torch.empty(U, dtype=torch.int32, device=dev)
        self.p_mode_words = torch.empty(U * K * W, dtype=torch.int64, device=dev)
        self.p_mode_count = torch.empty(U, dtype=torch.int32, device=dev)
        self.p_mci = torch.empty(U, dtype=torch.int32, device=dev)
        _lib.check(L.mchap_trace_posterior_batch_wph_device(
            U, self._p(self.d_units), self.S, self.Cn, int(burn), self._p(self.d_trace), int(max_states), K, W, self._p(self.p_words),
            self._p(self.p_counts), self._p(self.p_n), self._p(self.p_stats), self._p(self.p_mode), self._p(self.p_mode_words),
            self._p(self.p_mode_count), C.c_void_p(stream)))
        _lib.check(L.mchap_trace_incongruence_batch_wph_device(
            U, self._p(self.d_units), self.S, self.Cn, int(burn), self._p(self.d_trace), K, W, C.c_double(float(incongruence_threshold)),
            self._p(self.p_mci), C.c_void_p(stream)))
        self.burn = int(burn)
        self.incongruence_threshold = float(incongruence_threshold)
        self._end()

    def _summarise_listed(self, over):
        """The units `over` (int32 indices) summarised again with a table of chains x (steps - burn) states (as many as the LDS
        holds): their entries of p_n / p_stats / p_mode_words / p_mci are rewritten; returns (words int64 [len(over) * cap * K * wph],
        counts int32 [len(over) * cap] on the device, cap)."""
        torch = self.torch
        L = _lib.lib()
        K, W = self.Kmax, self.wph
        total = self.Cn * (self.S - self.burn)
        cap = min(total, int(L.mchap_trace_posterior_max_states_wph(K, W)))
        stream = torch.cuda.current_stream().cuda_stream
        d_list = torch.from_numpy(np.ascontiguousarray(over, dtype=np.int32)).to(self.device)
        o_words = torch.empty(len(over) * cap * K * W, dtype=torch.int64, device=self.device)
        o_counts = torch.empty(len(over) * cap, dtype=torch.int32, device=self.device)
        _lib.check(L.mchap_trace_posterior_listed_wph_device(
            len(over), self._p(d_list), self._p(self.d_units), self.S, self.Cn, self.burn, self._p(self.d_trace), cap, K, W,
            self._p(o_words), self._p(o_counts), self._p(self.p_n), self._p(self.p_stats), self._p(self.p_mode),
            self._p(self.p_mode_words), self._p(self.p_mode_count), C.c_void_p(stream)))
        _lib.check(L.mchap_trace_incongruence_listed_wph_device(
            len(over), self._p(d_list), self._p(self.d_units), self.S, self.Cn, self.burn, self._p(self.d_trace),
            min(self.S - self.burn, cap), K, W, C.c_double(self.incongruence_threshold), self._p(self.p_mci), C.c_void_p(stream)))
        return o_words, o_counts, cap

    def summary_arrays(self):
        """The posterior summaries of all units as arrays (what results() turns into one dict per unit): dict(n [U] distinct
        genotypes of each unit, plain [U] bool: the unit's summary is complete in these arrays -- the others (more distinct
        states than max_states, beyond a limit) go through results(only=...); words uint64 [N, K] / counts int32 [N]: the
        packed distinct genotypes of the plain units, unit after unit, most probable first (N = n[plain].sum(): only these
        rows leave the device; [N, K, 2] / [U, K, 2] for a batch of the general sampler: two words per haplotype); stats float64 [U, 2]
        (SPM, GPM), mode_words uint64 [U, K], mci [U], status [U], fixed int8
        flat (unit u at units_host['fixed_off'][u]), total = the steps a probability is a count of)."""
        torch = self.torch
        U, K, ms, Wp = self.n_units, self.Kmax, self.max_states, self.wph
        KW = K * Wp   # words of a state (a batch of the general sampler: two words per haplotype, the more significant first)
        self._begin()
        status, n, mci = self.d_status.cpu().numpy(), self.p_n.cpu().numpy(), self.p_mci.cpu().numpy()
        # units with more distinct genotypes than the batch kernels keep (samples with few or no reads: their chains wander):
        # summarised again by the listed launch, their rows taken from its table
        over = np.flatnonzero((status >= 0) & ((n < 0) | (n > ms) | (mci < 0))).astype(np.int32)
        cap = ms
        if len(over):
            o_words, o_counts, cap = self._summarise_listed(over)
            n, mci = self.p_n.cpu().numpy(), self.p_mci.cpu().numpy()
        wshape = (Wp,) if Wp > 1 else ()
        out = dict(n=n, stats=self.p_stats.cpu().numpy().reshape(U, 2),
                   mode_words=self.p_mode_words.cpu().numpy().view(np.uint64).reshape((U, K) + wshape),
                   mci=mci, status=status, fixed=self.d_fixed.cpu().numpy(), total=self.Cn * (self.S - self.burn))
        is_over = np.zeros(U, dtype=bool)
        is_over[over] = True
        out["plain"] = (status >= 0) & (n >= 0) & (n <= np.where(is_over, cap, ms)) & (mci >= 0)
        nn = np.where(out["plain"], n, 0).astype(np.int64)
        first = np.cumsum(nn) - nn                         # where each unit's rows start in the ragged result

        def gather(sel, table_row, words, counts):
            """rows of the units `sel` (indices): unit k's rows start at table_row[k] of the device table"""
            k_ = nn[sel]
            rows = np.repeat(table_row - (np.cumsum(k_) - k_), k_) + np.arange(int(k_.sum()), dtype=np.int64)
            d_rows = torch.from_numpy(rows).to(self.device)
            return words.view(-1, KW)[d_rows].cpu().numpy().view(np.uint64), counts[d_rows].cpu().numpy(), np.repeat(first[sel] - (np.cumsum(k_) - k_), k_) + np.arange(int(k_.sum()), dtype=np.int64)

        N = int(nn.sum())
        W, Cc = np.empty((N, KW), dtype=np.uint64), np.empty(N, dtype=np.int32)
        reg = np.flatnonzero(~is_over)
        w_, c_, dst = gather(reg, reg.astype(np.int64) * ms, self.p_words, self.p_counts)
        W[dst], Cc[dst] = w_, c_
        if len(over):
            w_, c_, dst = gather(over.astype(np.int64), np.arange(len(over), dtype=np.int64) * cap, o_words, o_counts)
            W[dst], Cc[dst] = w_, c_
        out["words"], out["counts"] = W.reshape((N, K) + wshape), Cc
        return out

    def results(self, raise_on_limit=True, only=None):
        """(only: the units to report, default all.)  Per unit: dict(genotypes int8 [n, K, M] distinct states (probability descending), probabilities [n], spm, gpm,
        mode_genotype int8 [K, M], mci, status).  Units with more distinct states than the batch kernels keep (512) are
        summarised by a second, listed launch with a table of chains x (steps - burn) states (as many as the LDS holds).
        A unit beyond the library's packed haplotype width raises NotImplementedError, or with raise_on_limit=False comes
        back as dict(status, limit=reason)."""
        U, K, ms, W = self.n_units, self.Kmax, self.max_states, self.wph
        wshape = (W,) if W > 1 else ()   # (a haplotype of a wide batch is two words: unpack_trace(..., W))
        self._begin()  # (the pass may have been issued on another stream)
        words = self.p_words.cpu().numpy().view(np.uint64).reshape((U, ms, K) + wshape)
        counts = self.p_counts.cpu().numpy().reshape(U, ms)
        n = self.p_n.cpu().numpy()
        stats = self.p_stats.cpu().numpy().reshape(U, 2)
        mode_words = self.p_mode_words.cpu().numpy().view(np.uint64).reshape((U, K) + wshape)
        mci = self.p_mci.cpu().numpy()
        status = self.d_status.cpu().numpy()
        fixed = self.d_fixed.cpu().numpy()
        total = self.Cn * (self.S - self.burn)
        # Units whose chains visited more distinct genotypes than the batch kernels keep (samples with few or no reads:
        # their chains wander): summarised again on the device with a table that cannot overflow, a list launch over
        # those units only -- no trace is downloaded, no posterior formed on the host
        over = np.flatnonzero((status >= 0) & ((n < 0) | (n > ms) | (mci < 0))).astype(np.int32)
        if only is not None:
            over = np.intersect1d(over, np.asarray(only, dtype=np.int32)).astype(np.int32)
        over_row = {}
        if len(over):
            o_words, o_counts, cap = self._summarise_listed(over)
            n = self.p_n.cpu().numpy()
            stats = self.p_stats.cpu().numpy().reshape(U, 2)
            mode_words = self.p_mode_words.cpu().numpy().view(np.uint64).reshape((U, K) + wshape)
            mci = self.p_mci.cpu().numpy()
            ow = o_words.cpu().numpy().view(np.uint64).reshape((len(over), cap, K) + wshape)
            oc = o_counts.cpu().numpy().reshape(len(over), cap)
            over_row = {int(u): (ow[i], oc[i], cap) for i, u in enumerate(over)}
        out = []
        wanted = range(U) if only is None else only
        host_cache = {"whole": 4 * len(wanted) >= U}  # (traces for _host_summary: the batch's at once, or a unit's slice at a time)
        for u in wanted:
            D = self.units_host[u]
            Ku, M, A = int(D["ploidy"]), int(D["n_pos"]), int(D["max_allele"])
            fx = fixed[int(D["fixed_off"]): int(D["fixed_off"]) + M]
            st = int(status[u])
            if st == _lib.UNIT_NAN_LLK:
                raise ValueError("Encountered log likelihood of nan")
            if st == _lib.UNIT_BREAKS:
                raise ValueError("breaks must be smaller then n")
            if st < 0:
                if raise_on_limit:
                    raise NotImplementedError("mchap_hip: unit %d exceeds the packed haplotype width" % u)
                out.append(dict(status=st, limit="more than %d bits of sampled alleles per haplotype (one bit per biallelic, two per "
                                                 "tri- / tetra-allelic SNV that is not fixed as homozygous)" % (128 if W > 1 else 64)))
                continue
            if u in over_row and 0 <= n[u] <= over_row[u][2] and mci[u] >= 0:
                w_, c_, _ = over_row[u]
                k = int(n[u])
                out.append(dict(genotypes=unpack_trace(w_[:k, :Ku], fx, A, W), probabilities=c_[:k] / total, spm=float(stats[u, 0]),
                                gpm=float(stats[u, 1]), mode_genotype=unpack_trace(mode_words[u, :Ku], fx, A, W), mci=int(mci[u]), status=st))
                continue
            if n[u] < 0 or n[u] > ms or mci[u] < 0:
                # (more distinct states than even the LDS holds -- tens of thousands of steps: the host classes on the trace)
                out.append(self._host_summary(u, st, fixed, host_cache))
                continue
            k = int(n[u])
            out.append(dict(genotypes=unpack_trace(words[u, :k, :Ku], fx, A, W), probabilities=counts[u, :k] / total,
                            spm=float(stats[u, 0]), gpm=float(stats[u, 1]), mode_genotype=unpack_trace(mode_words[u, :Ku], fx, A, W),
                            mci=int(mci[u]), status=st))
        return out
