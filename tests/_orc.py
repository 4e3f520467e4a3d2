"""ctypes binding of the CPU oracle (oracle/capi.cpp). Test infrastructure only."""
import ctypes
import numpy as np

vp = ctypes.c_void_p
c_ll = ctypes.c_longlong


def P(a):
    return a.ctypes.data_as(vp)


def load(so):
    L = ctypes.CDLL(so)
    L.orc_create.restype = vp
    L.orc_create.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_char_p]
    L.orc_last_error.restype = ctypes.c_char_p
    L.orc_bench_steps.restype = ctypes.c_double
    L.orc_bench_steps.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    L.orc_run_async.restype = ctypes.c_double
    L.orc_run_async.argtypes = [vp, ctypes.c_int, ctypes.c_double, vp]
    L.orc_destroy.argtypes = [vp]
    L.orc_init.argtypes = [vp, c_ll, ctypes.c_int, ctypes.c_int, vp, vp]
    L.orc_setup_chains.argtypes = [vp, c_ll, c_ll]
    L.orc_step.argtypes = [vp, ctypes.c_int]
    L.orc_film.argtypes = [vp, vp]
    L.orc_direct.argtypes = [vp, ctypes.c_int, vp]
    L.orc_stats.argtypes = [vp, vp]
    L.orc_info.argtypes = [vp, vp]
    L.orc_scene_params.argtypes = [vp, vp]
    L.orc_chain_summary.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int]
    L.orc_serialize_init_state.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int]
    L.orc_ref_eval.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    L.orc_trace.argtypes = [vp, ctypes.c_int, vp, vp, vp]
    L.orc_trace_brute.argtypes = [vp, ctypes.c_int, vp, vp, vp]
    L.orc_occluded.argtypes = [vp, ctypes.c_int, vp, vp]
    L.orc_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
    L.orc_pcg_u32.argtypes = [ctypes.c_ulonglong, ctypes.c_int, vp]
    L.orc_pcg_uniform.argtypes = [ctypes.c_ulonglong, ctypes.c_int, vp]
    L.orc_pcg_normal.argtypes = [ctypes.c_ulonglong, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp]
    L.orc_pcg_mixed.argtypes = [ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, vp]
    L.orc_pcg_dump.argtypes = [ctypes.c_ulonglong, ctypes.c_int, vp]
    L.orc_fastlog.argtypes = [ctypes.c_int, vp, vp]
    L.orc_kd_query.argtypes = [ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_float, ctypes.c_int, vp, vp, vp]
    L.orc_compute_gaussian.argtypes = [ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]
    return L


class Oracle:
    """Small convenience wrapper around one oracle scene instance."""

    def __init__(self, L, xml, force_diffuse=0, max_depth=0, width=0, height=0, seed_offset=-1, pathref=b""):
        self.L = L
        if isinstance(xml, str):
            xml = xml.encode()
        if isinstance(pathref, str):
            pathref = pathref.encode()
        h = L.orc_create(xml, force_diffuse, max_depth, width, height, seed_offset, pathref)
        if not h:
            raise RuntimeError(L.orc_last_error().decode())
        self.h = vp(h)
        info = (ctypes.c_int * 6)()
        L.orc_info(self.h, info)
        self.width, self.height, self.num_tris, self.max_depth, self.num_derv, self.num_lights = list(info)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def init(self, num_init, num_chains, init_threads):
        n = ctypes.c_float()
        nc = c_ll()
        if self.L.orc_init(self.h, num_init, num_chains, init_threads, ctypes.byref(n), ctypes.byref(nc)) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        self.num_chains = num_chains
        return n.value, nc.value

    def setup_chains(self, per_chain, extra=0):
        self.L.orc_setup_chains(self.h, per_chain, extra)

    def step(self, n):
        if self.L.orc_step(self.h, n) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())

    def run_async(self, threads, max_seconds=0.0):
        """the reference's scheduling: one chain per work item, immediate cache pushes; returns (chain-steps/s, steps done)"""
        done = c_ll()
        rate = self.L.orc_run_async(self.h, int(threads), float(max_seconds), ctypes.byref(done))
        return rate, done.value

    def film(self):
        f = np.zeros((self.height, self.width, 3), np.float32)
        self.L.orc_film(self.h, P(f))
        return f

    def direct(self, direct_spp):
        f = np.zeros((self.height, self.width, 3), np.float32)
        if self.L.orc_direct(self.h, int(direct_spp), P(f)) != 0:
            raise RuntimeError("orc_direct failed")
        return f

    def stats(self):
        s = (c_ll * 10)()
        self.L.orc_stats(self.h, s)
        keys = ["steps", "largeSteps", "accepted", "gradCalls", "cacheQueries", "cacheHits", "resets", "cacheReadyMask"]
        d = dict(zip(keys, list(s)[:8]))
        d["weightSum"] = ctypes.c_double.from_buffer(s, 8 * 8).value
        return d

    def summary(self, which=0):
        out = np.zeros((self.num_chains, 32), np.float32)
        self.L.orc_chain_summary(self.h, which, P(out), 32)
        return out

    def scene_params(self):
        s = np.zeros(38, np.float32)
        self.L.orc_scene_params(self.h, P(s))
        return s

    def serialize_init_state(self, i):
        prim = np.zeros(17, np.float32)
        vert = np.zeros(1000, np.float32)
        r = self.L.orc_serialize_init_state(self.h, i, P(prim), 17, P(vert), 1000)
        if r < 0:
            return None
        return r >> 4, r & 15, prim, vert

    def ref_eval(self, c, l, prim, vert):
        ll = np.zeros(1, np.float32)
        g = np.zeros(16, np.float32)
        r = self.L.orc_ref_eval(self.h, c, l, P(prim), P(vert), P(ll), P(g))
        if r != 0:
            return None
        return ll[0], g[: 2 * max(c + l - 1, 2)]
