"""ctypes binding of include/spartan_hip.h. Raises if the HIP library is missing (no fallback)."""
import ctypes, os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPARTAN_HIP_LIB") or os.path.join(_HERE, "lib", "libspartan_hip.so")  # the override is for diagnostic builds (bench/ktime_probe.py)
sz = ctypes.c_size_t
vp = ctypes.c_void_p
u64p = ctypes.POINTER(ctypes.c_uint64)


class SpartanHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise SpartanHipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    L = ctypes.CDLL(LIB_PATH)
    L.sp_strerror.restype = ctypes.c_char_p
    L.sp_version.restype = ctypes.c_char_p
    L.sp_gens_len.restype = sz
    L.sp_table_len.restype = sz
    L.sp_gens_len.argtypes = [vp]
    L.sp_gens_table_bytes.restype = sz
    L.sp_gens_table_bytes.argtypes = [vp]
    L.sp_ctx_trips.restype = ctypes.c_uint64
    L.sp_ctx_trips.argtypes = [vp]
    L.sp_ctx_device.argtypes = [vp]
    L.sp_host_zk_ahead_free.argtypes = [vp]
    L.sp_table_len.argtypes = [vp]
    L.sp_gens_free.argtypes = [vp]
    L.sp_table_free.argtypes = [vp]
    L.sp_ctx_destroy.argtypes = [vp]
    return L


lib = _load()


def set_default_option(key, value):
    """process-wide default of a library option: contexts created afterwards start from it (sp_ctx_set_option with a NULL context)"""
    _chk(lib.sp_ctx_set_option(None, key.encode(), str(int(value)).encode()))


def options_table():
    """[(key, default, min, max, tier, doc)] of the library's option table (sp_option_describe)"""
    out, i = [], 0
    while True:
        k = ctypes.c_char_p(); d = ctypes.c_int64(); lo = ctypes.c_int64(); hi = ctypes.c_int64(); t = ctypes.c_int(); doc = ctypes.c_char_p()
        if lib.sp_option_describe(ctypes.c_int(i), ctypes.byref(k), ctypes.byref(d), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(t), ctypes.byref(doc)) != 0:
            return out
        out.append((k.value.decode(), d.value, lo.value, hi.value, t.value, doc.value.decode()))
        i += 1

# every symbol include/spartan_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = ["sp_ctx_set_option", "sp_ctx_get_option", "sp_ctx_copy_options", "sp_option_describe", "sp_host_msm2_probe", "sp_hash_layer_first", "sp_product_tree_many_from", "sp_sumcheck_eval_batched_eq", "sp_sumcheck_bind_eval_batched_eq", "sp_table_scale_prefix", "sp_sparse_entry_index", "sp_sparse_entry_values", "sp_tables_pack", "sp_tables_unpack_residues", "sp_table_residue_split", "sp_table_set_len", "sp_table_add_into", "sp_ctx_device", "sp_ctx_trips", "sp_gens_table_bytes", "sp_host_commit_small", "sp_host_commit_point", "sp_host_commit_probe",
           "sp_host_zk_ahead_begin", "sp_host_zk_ahead_wait", "sp_host_zk_ahead_free", "sp_addr_timestamps", "sp_commit_rows", "sp_commit_rows_dev", "sp_commit_rows_dev_begin", "sp_commit_rows_dev_start", "sp_commit_rows_partial", "sp_host_points_sum_encode", "sp_commit_rows_upload_start", "sp_job_wait", "sp_ctx_create", "sp_ctx_destroy", "sp_ctx_sync", "sp_dot", "sp_dot3", "sp_dot3_many", "sp_dot_many", "sp_eq_expand", "sp_evaluate", "sp_gather", "sp_gens_free", "sp_gens_from_uniform", "sp_gens_len", "sp_gens_window_bits", "sp_gens_windows", "sp_gens_plan_pair", "sp_gens_upload", "sp_hash_layer", "sp_index_free", "sp_index_upload", "sp_ipa_begin", "sp_ipa_begin_dev", "sp_ipa_set_scale", "sp_ipa_commit_ghat", "sp_ipa_finish", "sp_ipa_finish_commit", "sp_ipa_free", "sp_ipa_round_fold", "sp_ipa_round_lr", "sp_ipa_round_prelaunch", "sp_msm_indexed", "sp_product_tree", "sp_product_tree_many", "sp_prof_enable", "sp_prof_read", "sp_prof_read_ops", "sp_prof_read_shapes", "sp_prof_read_spans", "sp_msm_window_bits", "sp_prof_reset", "sp_prof_select", "sp_sparse_eval_table", "sp_sparse_evaluate", "sp_sparse_evaluate_begin", "sp_sparse_free", "sp_sparse_mulvec", "sp_sparse_upload", "sp_strerror", "sp_sumcheck_bind_eval", "sp_sumcheck_bind_eval_start", "sp_sumcheck_bind_eval_collect", "sp_sumcheck_bind_eval_commit", "sp_sumcheck_bind_eval_batched", "sp_sumcheck_eval", "sp_sumcheck_eval_batched", "sp_sumcheck_eval_coeffs_batched", "sp_sumcheck_bind2_eval_batched", "sp_sumcheck_bind2_eval_tables_batched", "sp_table_alloc", "sp_table_alloc_uninit", "sp_table_bind_top", "sp_table_bind_top_heads", "sp_table_clone", "sp_table_copy", "sp_table_download", "sp_table_free", "sp_table_from_index", "sp_table_heads", "sp_table_gather", "sp_table_len", "sp_table_upload", "sp_table_view", "sp_table_write", "sp_vecmat", "sp_vecmat_dev", "sp_vecmat_tab", "sp_version"]


def _chk(rc):
    if rc != 0:
        raise SpartanHipError(f"spartan_hip error {rc}: {lib.sp_strerror(rc).decode()}")


def _u64(buf):
    """bytes/bytearray/ctypes array of Montgomery limbs -> ctypes pointer"""
    if isinstance(buf, (bytes, bytearray)):
        return ctypes.cast(ctypes.create_string_buffer(bytes(buf), len(buf)), u64p)
    return ctypes.cast(buf, u64p)


class Ctx:
    def __init__(self, device=0):
        self.h = vp()
        _chk(lib.sp_ctx_create(ctypes.c_int(device), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib.sp_ctx_destroy(self.h)
            self.h = vp()

    def set_option(self, key, value):
        """a library option of this context (include/spartan_hip.h, sp_ctx_set_option; table: spartan_amd/csrc/options.hpp)"""
        _chk(lib.sp_ctx_set_option(self.h, key.encode(), str(int(value)).encode()))

    def get_option(self, key):
        v = ctypes.c_int64()
        _chk(lib.sp_ctx_get_option(self.h, key.encode(), ctypes.byref(v)))
        return v.value

    def prof_enable(self, on=True):
        _chk(lib.sp_prof_enable(self.h, ctypes.c_int(1 if on else 0)))

    def prof_reset(self):
        _chk(lib.sp_prof_reset(self.h))

    def prof_read(self):
        cap = 64
        names = (ctypes.c_char_p * cap)(); ms = (ctypes.c_double * cap)(); n = (ctypes.c_uint64 * cap)(); by = (ctypes.c_double * cap)()
        k = lib.sp_prof_read(self.h, names, ms, n, by, ctypes.c_int(cap))
        return {names[i].decode(): {"ms": ms[i], "launches": int(n[i]), "alg_bytes": by[i]} for i in range(min(k, cap))}


class Gens:
    def __init__(self, ctx, compressed=None, uniform=None):
        self.ctx = ctx
        self.h = vp()
        if compressed is not None:
            n = len(compressed) // 32
            _chk(lib.sp_gens_upload(ctx.h, compressed, sz(n), ctypes.byref(self.h)))
            self.compressed = bytes(compressed)
        else:
            n = len(uniform) // 64
            out = (ctypes.c_uint8 * (32 * n))()
            _chk(lib.sp_gens_from_uniform(ctx.h, uniform, sz(n), out, ctypes.byref(self.h)))
            self.compressed = bytes(out)
        self.n = n

    def window_bits(self):
        return int(lib.sp_gens_window_bits(self.h))

    def windows(self):
        """windows per scalar = mixed additions per committed scalar of this set's tables"""
        return int(lib.sp_gens_windows(self.h))

    def free(self):
        if self.h:
            lib.sp_gens_free(self.h)
            self.h = vp()

    def commit_rows(self, Z, rows, cols, blinds=None, g_off=0, h_idx=None):
        out = (ctypes.c_uint8 * (32 * rows))()
        if h_idx is None:
            h_idx = self.n - 1
        if isinstance(Z, Table):
            _chk(lib.sp_commit_rows_dev(self.ctx.h, self.h, sz(g_off), sz(h_idx), Z.h, sz(0), sz(rows), sz(cols),
                                        _u64(blinds) if blinds is not None else None, out))
        else:
            _chk(lib.sp_commit_rows(self.ctx.h, self.h, sz(g_off), sz(h_idx), _u64(Z), sz(rows), sz(cols),
                                    _u64(blinds) if blinds is not None else None, out))
        return bytes(out)

    def commit_rows_begin(self, Z, rows, cols, g_off=0, z_off=0):
        """background-stream commit (no blinds); returns a job handle for commit_rows_wait"""
        job = vp()
        _chk(lib.sp_commit_rows_dev_begin(self.ctx.h, self.h, sz(g_off), Z.h, sz(z_off), sz(rows), sz(cols), ctypes.byref(job)))
        return (job, rows)

    def commit_rows_wait(self, job):
        out = (ctypes.c_uint8 * (32 * job[1]))()
        _chk(lib.sp_job_wait(job[0], out))
        return bytes(out)

    def msm_indexed(self, idx, S, rows=1):
        cols = len(idx)
        arr = (ctypes.c_uint32 * cols)(*idx)
        out = (ctypes.c_uint8 * (32 * rows))()
        _chk(lib.sp_msm_indexed(self.ctx.h, self.h, arr, sz(cols), _u64(S), sz(rows), out))
        return bytes(out)


class Table:
    def __init__(self, ctx, h):
        self.ctx = ctx
        self.h = h

    @staticmethod
    def upload(ctx, Z, n):
        h = vp()
        _chk(lib.sp_table_upload(ctx.h, _u64(Z), sz(n), ctypes.byref(h)))
        return Table(ctx, h)

    @staticmethod
    def alloc(ctx, n):
        """zero-filled table of n scalars"""
        h = vp()
        _chk(lib.sp_table_alloc(ctx.h, sz(n), ctypes.byref(h)))
        return Table(ctx, h)

    @staticmethod
    def eq(ctx, r, ell):
        h = vp()
        _chk(lib.sp_eq_expand(ctx.h, _u64(r), sz(ell), ctypes.byref(h)))
        return Table(ctx, h)

    def __len__(self):
        return lib.sp_table_len(self.h)

    def download(self, n=None, off=0):
        n = len(self) if n is None else n
        out = (ctypes.c_uint64 * (4 * n))()
        _chk(lib.sp_table_download(self.ctx.h, self.h, sz(off), sz(n), out))
        return out

    def free(self):
        if self.h:
            lib.sp_table_free(self.h)
            self.h = vp()


def sumcheck_eval(ctx, kind, tabs):
    arr = (vp * len(tabs))(*[t.h for t in tabs])
    out = (ctypes.c_uint64 * 12)()
    _chk(lib.sp_sumcheck_eval(ctx.h, ctypes.c_int(kind), arr, sz(len(tabs)), out))
    return out


def sumcheck_bind_eval(ctx, kind, tabs, r):
    arr = (vp * len(tabs))(*[t.h for t in tabs])
    out = (ctypes.c_uint64 * 12)()
    _chk(lib.sp_sumcheck_bind_eval(ctx.h, ctypes.c_int(kind), arr, sz(len(tabs)), _u64(r), out))
    return out


def sumcheck_bind_eval_start(ctx, kind, tabs, r):
    arr = (vp * len(tabs))(*[t.h for t in tabs])
    _chk(lib.sp_sumcheck_bind_eval_start(ctx.h, ctypes.c_int(kind), arr, sz(len(tabs)), _u64(r)))


def sumcheck_bind_eval_collect(ctx):
    out = (ctypes.c_uint64 * 12)()
    _chk(lib.sp_sumcheck_bind_eval_collect(ctx.h, out))
    return out


def bind_top(ctx, tabs, r):
    arr = (vp * len(tabs))(*[t.h for t in tabs])
    _chk(lib.sp_table_bind_top(ctx.h, arr, sz(len(tabs)), _u64(r)))


def vecmat(ctx, L, Lsz, Z):
    R = len(Z) // Lsz
    out = (ctypes.c_uint64 * (4 * R))()
    _chk(lib.sp_vecmat(ctx.h, _u64(L), sz(Lsz), Z.h, out))
    return out


def dot(ctx, a, b, n, a_off=0, b_off=0):
    out = (ctypes.c_uint64 * 4)()
    _chk(lib.sp_dot(ctx.h, a.h, sz(a_off), b.h, sz(b_off), sz(n), out))
    return out


def evaluate(ctx, Z, r, ell):
    out = (ctypes.c_uint64 * 4)()
    _chk(lib.sp_evaluate(ctx.h, Z.h, _u64(r), sz(ell), out))
    return out


def heads(ctx, tabs):
    arr = (vp * len(tabs))(*[t.h for t in tabs])
    out = (ctypes.c_uint64 * (4 * len(tabs)))()
    _chk(lib.sp_table_heads(ctx.h, arr, sz(len(tabs)), out))
    return out
