"""ctypes binding of the MI355X back end (liblmc_hip.so, C ABI in include/lmc_abi.h).

The package is plumbing around the C ABI: it loads the HIP library (and fails loudly if it is missing or if no GPU is
usable); it never computes anything on the CPU.  Import with importlib.import_module("langevin-mcmc_amd")."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("LMC_LIB") or os.path.join(_HERE, "liblmc_hip.so")  # LMC_LIB: A/B builds of the same library (scripts/)
vp = ctypes.c_void_p
c_ll = ctypes.c_longlong


class SceneDesc(ctypes.Structure):
    _fields_ = [
        ("scene_xml", ctypes.c_char_p),
        ("force_diffuse", ctypes.c_int),
        ("max_depth", ctypes.c_int),
        ("width", ctypes.c_int),
        ("height", ctypes.c_int),
        ("seed_offset", ctypes.c_int),
        ("device", ctypes.c_int),
        ("use_gradient", ctypes.c_int),
    ]


def P(a):
    return a.ctypes.data_as(vp)


_lib = None


def lib():
    """The loaded HIP library; raises if it has not been built (run `python __graft_entry__.py`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP extension missing: %s (build with `python __graft_entry__.py`); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.lmc_last_error.restype = ctypes.c_char_p
        L.lmc_create.restype = vp
        L.lmc_create.argtypes = [ctypes.POINTER(SceneDesc)]
        L.lmc_destroy.argtypes = [vp]
        L.lmc_info.argtypes = [vp, vp]
        L.lmc_scene_params.argtypes = [vp, vp]
        L.lmc_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
        L.lmc_chains_init.argtypes = [vp, c_ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_ll, c_ll]
        L.lmc_init_result.argtypes = [vp, vp, vp]
        L.lmc_chains_step.argtypes = [vp, ctypes.c_int]
        L.lmc_sync.argtypes = [vp]
        L.lmc_film_read.argtypes = [vp, vp]
        L.lmc_film_clear.argtypes = [vp]
        L.lmc_stats.argtypes = [vp, vp, vp]
        L.lmc_chain_summary.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int]
        L.lmc_step_timing.argtypes = [vp, vp, vp]
        L.lmc_relocation_stats.argtypes = [vp, vp]
        if hasattr(L, "lmc_relocation_skipped"):  # (an older A/B build selected with LMC_LIB lacks it)
            L.lmc_relocation_skipped.argtypes = [vp]
            L.lmc_relocation_skipped.restype = c_ll
        L.lmc_kernel_timing.argtypes = [vp, vp]
        L.lmc_kernel_timing_split.argtypes = [vp, vp]
        L.lmc_get_option.argtypes = [vp, ctypes.c_char_p, vp]
        L.lmc_output_name.argtypes = [vp]
        L.lmc_output_name.restype = ctypes.c_char_p
        L.lmc_image_read.argtypes = [ctypes.c_char_p, vp, vp, vp]
        L.lmc_image_write_exr.argtypes = [ctypes.c_char_p, vp, ctypes.c_int, ctypes.c_int]
        L.lmc_direct_lighting.argtypes = [vp, ctypes.c_int]
        L.lmc_direct_read.argtypes = [vp, vp]
        L.lmc_path_trace.argtypes = [vp, ctypes.c_int]
        L.lmc_bidir_mc.argtypes = [vp, ctypes.c_int]
        L.lmc_stream_probe.argtypes = [c_ll, ctypes.c_int]
        L.lmc_grad_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
        L.lmc_trace.argtypes = [vp, ctypes.c_int, vp, vp, vp]
        L.lmc_occluded.argtypes = [vp, ctypes.c_int, vp, vp]
        L.lmc_rng_probe.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp]
        L.lmc_kd_probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_float, ctypes.c_int, vp, vp, vp]
        L.lmc_gauss_probe.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        L.lmc_comm_unique_id.argtypes = [vp]
        L.lmc_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
        L.lmc_film_allreduce.argtypes = [vp]
        L.lmc_film_device_ptr.argtypes = [vp, vp]
        L.lmc_film_device_ptr.restype = vp
        L.lmc_comm_allreduce_f64.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
        L.lmc_comm_barrier.argtypes = [vp]
        L.lmc_host_issue_timing.argtypes = [vp, vp, vp]
        L.lmc_group_info.argtypes = [vp, ctypes.c_int, vp]
        _lib = L
    return _lib


def _err():
    return lib().lmc_last_error().decode()


class Renderer:
    """One scene resident on one GPU; mirrors MLT() of the reference (mlt.cpp:20-214) step by step."""

    def __init__(self, xml, force_diffuse=0, max_depth=0, width=0, height=0, seed_offset=-1, device=0, use_gradient=1):
        L = lib()
        self._xml = os.fsencode(xml)
        d = SceneDesc(self._xml, force_diffuse, max_depth, width, height, seed_offset, device, use_gradient)
        h = L.lmc_create(ctypes.byref(d))
        if not h:
            raise RuntimeError("lmc_create failed: " + _err())
        self.h = vp(h)
        info = (ctypes.c_int * 8)()
        L.lmc_info(self.h, info)
        (self.width, self.height, self.num_tris, self.max_depth, self.num_nodes, self.bvh_depth, self.num_lights, self.mala) = list(info)
        self.num_chains = 0

    def close(self):
        if self.h:
            lib().lmc_destroy(self.h)
            self.h = None

    def set_option(self, name, value):
        if lib().lmc_set_option(self.h, name.encode(), float(value)) != 0:
            raise RuntimeError(_err())

    def get_option(self, name):
        v = ctypes.c_double()
        if lib().lmc_get_option(self.h, name.encode(), ctypes.byref(v)) != 0:
            raise RuntimeError(_err())
        return v.value

    def scene_params(self):
        s = np.zeros(38, np.float32)
        lib().lmc_scene_params(self.h, P(s))
        return s

    def init_chains(self, num_init, n_chains_total, init_threads, per_chain, extra=0, chain_begin=0, chain_end=None):
        if chain_end is None:
            chain_end = n_chains_total
        if lib().lmc_chains_init(self.h, num_init, n_chains_total, init_threads, chain_begin, chain_end, per_chain, extra) != 0:
            raise RuntimeError("lmc_chains_init failed: " + _err())
        self.num_chains = chain_end - chain_begin
        self.num_chains_total = n_chains_total
        n = ctypes.c_float()
        nc = c_ll()
        lib().lmc_init_result(self.h, ctypes.byref(n), ctypes.byref(nc))
        self.normalization = n.value
        return n.value, nc.value

    def step(self, n):
        if lib().lmc_chains_step(self.h, n) != 0:
            raise RuntimeError("lmc_chains_step failed: " + _err())

    def sync(self):
        if lib().lmc_sync(self.h) != 0:
            raise RuntimeError(_err())

    def film(self):
        f = np.zeros((self.height, self.width, 3), np.float32)
        if lib().lmc_film_read(self.h, P(f)) != 0:
            raise RuntimeError(_err())
        return f

    # ---- multi-GPU (include/lmc_abi.h): in-library RCCL sum of the device films
    def comm_init(self, n_ranks, rank, id128):
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(id128))
        if lib().lmc_comm_init(self.h, n_ranks, rank, buf) != 0:
            raise RuntimeError("lmc_comm_init failed: " + _err())

    def film_allreduce(self):
        if lib().lmc_film_allreduce(self.h) != 0:
            raise RuntimeError("lmc_film_allreduce failed: " + _err())

    def comm_allreduce(self, values, op="sum"):
        """host doubles reduced over the ranks of the job's communicator (op: sum / max / min); returns a list"""
        v = (ctypes.c_double * len(values))(*[float(x) for x in values])
        if lib().lmc_comm_allreduce_f64(self.h, v, len(values), {"sum": 0, "max": 1, "min": 2}[op]) != 0:
            raise RuntimeError("lmc_comm_allreduce_f64 failed: " + _err())
        return list(v)

    def comm_gather(self, value, rank, world):
        """every rank's scalar, in rank order, on every rank (a sum of one-hot vectors over the job's communicator)"""
        return self.comm_allreduce([float(value) if k == rank else 0.0 for k in range(world)], "sum")

    def comm_barrier(self):
        if lib().lmc_comm_barrier(self.h) != 0:
            raise RuntimeError("lmc_comm_barrier failed: " + _err())

    def host_issue_timing(self):
        """(ms of host time spent queueing this context's steps since the last call, steps covered)"""
        ms, n = ctypes.c_double(), c_ll()
        if lib().lmc_host_issue_timing(self.h, ctypes.byref(ms), ctypes.byref(n)) != 0:
            raise RuntimeError(_err())
        return ms.value, n.value

    def direct_lighting(self, direct_spp):
        """DirectLighting pre-pass (direct.cpp); returns the un-normalised direct buffer [H, W, 3]."""
        if lib().lmc_direct_lighting(self.h, int(direct_spp)) != 0:
            raise RuntimeError(_err())
        out = np.zeros((self.height, self.width, 3), np.float32)
        if lib().lmc_direct_read(self.h, P(out)) != 0:
            raise RuntimeError(_err())
        return out

    def path_trace(self, spp):
        """Unidirectional path tracing with next-event estimation over the scene's full depth range (cross-check estimator)."""
        if lib().lmc_path_trace(self.h, int(spp)) != 0:
            raise RuntimeError(_err())
        out = np.zeros((self.height, self.width, 3), np.float32)
        if lib().lmc_direct_read(self.h, P(out)) != 0:
            raise RuntimeError(_err())
        return out

    def bidir_mc(self, spp):
        """Plain Monte Carlo over bidirectional samples (path length >= 3): radiance image [H, W, 3]."""
        if lib().lmc_bidir_mc(self.h, int(spp)) != 0:
            raise RuntimeError(_err())
        out = np.zeros((self.height, self.width, 3), np.float32)
        if lib().lmc_direct_read(self.h, P(out)) != 0:
            raise RuntimeError(_err())
        return out

    def stats(self):
        s = (c_ll * 8)()
        w = ctypes.c_double()
        if lib().lmc_stats(self.h, s, ctypes.byref(w)) != 0:
            raise RuntimeError(_err())
        keys = ["steps", "largeSteps", "accepted", "gradCalls", "cacheQueries", "cacheHits", "resets", "cacheReadyMask"]
        d = dict(zip(keys, list(s)))
        d["weightSum"] = w.value
        return d

    def relocation_stats(self):
        """None when chain relocation is off; else relocations run, chains moved by the last one, technique breaks between adjacent slots, slots"""
        o = (c_ll * 4)()
        r = lib().lmc_relocation_stats(self.h, o)
        if r == -1:
            return None
        if r != 0:
            raise RuntimeError(_err())
        return dict(relocations=o[0], moved=o[1], breaks=o[2], slots=o[3], skipped=int(lib().lmc_relocation_skipped(self.h)) if hasattr(lib(), "lmc_relocation_skipped") else 0)

    def summary(self, which=0):
        n = self.num_chains  # which = 0: current states, 1: init states -- of this rank's chains
        out = np.zeros((n, 32), np.float32)
        if lib().lmc_chain_summary(self.h, which, P(out), 32) < 0:
            raise RuntimeError(_err())
        return out

    def step_timing(self):
        ms = ctypes.c_double()
        n = c_ll()
        if lib().lmc_step_timing(self.h, ctypes.byref(ms), ctypes.byref(n)) != 0:
            raise RuntimeError(_err())
        return ms.value, n.value

    def kernel_timing(self):
        """(ms in the lean small-step kernel, ms in the large/generic launches) over the interval of the last
        step_timing() call, and the cumulative number of chain-steps the lean kernel has run."""
        out = (ctypes.c_double * 3)()
        if lib().lmc_kernel_timing(self.h, out) != 0:
            raise RuntimeError(_err())
        return out[0], out[1], int(out[2])

    def kernel_timing_split(self):
        """the three step launches separately over the interval of the last step_timing() call: ms in the lean small-step kernel,
        in the large-step launch and in the generic small-step launch (cache-filling gradient steps / every H2MC small step), plus
        the cumulative number of chain-steps the lean kernel has run"""
        out = (ctypes.c_double * 4)()
        if lib().lmc_kernel_timing_split(self.h, out) != 0:
            raise RuntimeError(_err())
        return {"lean_ms": out[0], "large_ms": out[1], "generic_ms": out[2], "lean_steps": int(out[3])}

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        n = len(rays)
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        if lib().lmc_trace(self.h, n, P(rays), P(prim), P(t)) != 0:
            raise RuntimeError(_err())
        return prim, t

    def occluded(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        occ = np.zeros(len(rays), np.int32)
        if lib().lmc_occluded(self.h, len(rays), P(rays), P(occ)) != 0:
            raise RuntimeError(_err())
        return occ


class Group:
    """In-process ranks: the given Renderers (one per GPU, or several on one GPU for bring-up / tests) become ranks 0 .. n-1 of ONE job
    (lmc_group_chains_init / lmc_group_chains_step): MLTInit sharded by init stream, chains split into contiguous equal ranges, the
    global cache's pushes exchanged every step -- the same trajectories as a single rank holding all the chains."""

    def __init__(self, renderers):
        self.rens = list(renderers)
        self._arr = (vp * len(self.rens))(*[r.h for r in self.rens])

    def init_chains(self, num_init, n_chains_total, init_threads, per_chain, extra=0):
        L = lib()
        L.lmc_group_chains_init.argtypes = [vp, ctypes.c_int, c_ll, ctypes.c_int, ctypes.c_int, c_ll, c_ll]
        if L.lmc_group_chains_init(self._arr, len(self.rens), num_init, n_chains_total, init_threads, per_chain, extra) != 0:
            raise RuntimeError("lmc_group_chains_init failed: " + _err())
        from . import sharding

        for ren, (b, e) in zip(self.rens, sharding.group_ranges(n_chains_total, len(self.rens))):
            ren.num_chains, ren.num_chains_total = e - b, n_chains_total
            nn, nc = ctypes.c_float(), c_ll()
            L.lmc_init_result(ren.h, ctypes.byref(nn), ctypes.byref(nc))
            ren.normalization = nn.value
        return self.rens[0].normalization, nc.value

    def step(self, n):
        L = lib()
        L.lmc_group_chains_step.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        if L.lmc_group_chains_step(self._arr, len(self.rens), n) != 0:
            raise RuntimeError("lmc_group_chains_step failed: " + _err())

    def info(self):
        """distinct devices, ordered device pairs, pairs with direct peer access enabled, host threads driving the steps"""
        o = (c_ll * 4)()
        if lib().lmc_group_info(self._arr, len(self.rens), o) != 0:
            raise RuntimeError(_err())
        return dict(devices=o[0], peer_pairs=o[1], peer_pairs_enabled=o[2], host_threads=o[3])

    def film_reduce(self):
        """Every member's device film becomes the sum over the members (peer copies: the in-process lmc_film_allreduce); returns its wall time in ms."""
        L = lib()
        L.lmc_group_film_reduce.argtypes = [vp, ctypes.c_int, vp]
        ms = ctypes.c_double()
        if L.lmc_group_film_reduce(self._arr, len(self.rens), ctypes.byref(ms)) != 0:
            raise RuntimeError("lmc_group_film_reduce failed: " + _err())
        return ms.value


def device_count():
    """HIP devices visible to this process (0 without a GPU)."""
    return int(lib().lmc_device_count())


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it, the host program broadcasts it)"""
    buf = (ctypes.c_ubyte * 128)()
    if lib().lmc_comm_unique_id(buf) != 0:
        raise RuntimeError("lmc_comm_unique_id failed: " + _err())
    return bytes(buf)


def read_image(path):
    """EXR / PNG -> float32 [H, W, 3] through the library's own readers."""
    w, h = ctypes.c_int(), ctypes.c_int()
    if lib().lmc_image_read(path.encode(), ctypes.byref(w), ctypes.byref(h), None) != 0:
        raise RuntimeError(_err())
    out = np.zeros((h.value, w.value, 3), np.float32)
    if lib().lmc_image_read(path.encode(), None, None, P(out)) != 0:
        raise RuntimeError(_err())
    return out


def write_exr(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    if lib().lmc_image_write_exr(path.encode(), P(rgb), rgb.shape[1], rgb.shape[0]) != 0:
        raise RuntimeError(_err())


def grad_batch(c, l, primary_soa, scene38, vert_soa, want_grad=True):
    """n evaluations of the (c,l) path program on the GPU; SoA word-major inputs (see include/lmc_abi.h)."""
    primary_soa = np.ascontiguousarray(primary_soa, np.float32)
    vert_soa = np.ascontiguousarray(vert_soa, np.float32)
    scene38 = np.ascontiguousarray(scene38, np.float32)
    n = primary_soa.shape[1]
    L = max(c + l - 1, 2)
    ll = np.zeros(n, np.float32)
    g = np.zeros((2 * L, n), np.float32)
    r = lib().lmc_grad_batch(c, l, n, P(primary_soa), P(scene38), P(vert_soa), P(ll), P(g) if want_grad else None)
    if r != 0:
        raise RuntimeError("lmc_grad_batch failed: " + _err())
    return ll, g


def smoke():
    """One small invocation of the hot path on cuda:0, checked against the CPU oracle (test infrastructure)."""
    import sys

    sys.path.insert(0, ROOT)
    from tests import gpu_checks

    gpu_checks.smoke()
