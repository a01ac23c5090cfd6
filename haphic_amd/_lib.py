"""ctypes binding of libhaphic_hip.so (include/haphic_hip.h).

This is the stub a HapHiC maintainer would add next to scripts/HapHiC_cluster.py.
There is NO CPU fallback: if the HIP library is missing or no GPU is visible, loading / the first
call raises, exactly as a missing sparse_dot_mkl would — but loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('HAPHIC_HIP_SO') or os.path.join(_HERE, 'libhaphic_hip.so')      # (the override: the sanitizer build of build.build_asan)

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_vpp = C.POINTER(C.c_void_p)


class IngestConfig(C.Structure):
    _fields_ = [('n_ctg', C.c_int32), ('n_frag', C.c_int32),
                ('ctg_rank', c_i32p), ('ctg_len', c_i64p), ('ctg_frag0', c_i32p), ('ctg_split', c_u8p),
                ('frag_rank', c_i32p), ('frag_len', c_i64p), ('frag_nx', c_u8p),
                ('bin_size', C.c_int64), ('flank', C.c_int64), ('bins', C.c_int32), ('skip_intra', C.c_int32),
                ('expected_keys', C.c_int64)]


# name -> (restype, argtypes); every symbol include/haphic_hip.h declares
SIGNATURES = {
    'hhx_last_error': (C.c_char_p, []),
    'hhx_version': (C.c_int, []),
    'hhx_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'hhx_set_device': (C.c_int, [C.c_int]),
    'hhx_set_stream': (C.c_int, [C.c_void_p]),
    'hhx_synchronize': (C.c_int, []),
    'hhx_pool_trim': (C.c_int, []),
    'hhx_pool_trim_keep': (C.c_int, [C.c_int64]),
    'hhx_pool_prewarm': (C.c_int, [C.c_int32, c_i64p]),
    'hhx_tune': (C.c_int, [C.c_char_p, C.c_int64]),
    'hhx_profile_enable': (C.c_int, [C.c_int]),
    'hhx_profile_reset': (C.c_int, []),
    'hhx_profile_get': (C.c_int, [C.c_char_p, c_f64p, c_i64p]),
    'hhx_profile_counter': (C.c_int, [C.c_char_p, c_i64p]),
    'hhx_csr_from_host': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, c_vpp]),
    'hhx_csr_from_device': (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, c_vpp]),
    'hhx_csr_shape': (C.c_int, [C.c_void_p, c_i32p, c_i32p, c_i64p]),
    'hhx_csr_to_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_csr_device_ptrs': (C.c_int, [C.c_void_p, c_vpp, c_vpp, c_vpp]),
    'hhx_csr_copy': (C.c_int, [C.c_void_p, c_vpp]),
    'hhx_csr_row_block': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_vpp]),
    'hhx_csr_free': (C.c_int, [C.c_void_p]),
    'hhx_normalize_l1': (C.c_int, [C.c_void_p]),
    'hhx_inflate': (C.c_int, [C.c_void_p, C.c_double]),
    'hhx_prune': (C.c_int, [C.c_void_p, C.c_double, c_vpp]),
    'hhx_inflate_prune': (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_vpp]),
    'hhx_spgemm': (C.c_int, [C.c_void_p, C.c_void_p, c_vpp]),
    'hhx_spgemm_ex': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_vpp, c_i64p]),
    'hhx_expand_inflate_prune': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, c_vpp, c_i64p, c_i64p]),
    'hhx_link_weights': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_double, c_i64p]),
    'hhx_group_link_sums': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'hhx_row_products': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_expand_links': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_double, C.c_double, c_vpp, c_i64p, c_i64p]),
    'hhx_convergence_stat': (C.c_int, [C.c_void_p, C.c_void_p, c_f32p]),
    'hhx_mcl': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, c_vpp, C.POINTER(C.c_int),
                          C.POINTER(C.c_int), C.c_void_p]),
    'hhx_mcl_normalized': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, c_vpp, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.c_void_p]),
    'hhx_mcl_links': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, c_vpp, C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.c_void_p]),
    'hhx_interpret': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i32p]),
    'hhx_dict_to_matrix': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_void_p,
                                     C.c_int32, C.c_int, C.c_void_p, c_i32p, c_vpp]),
    'hhx_count_re_sites': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_rank_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'hhx_ingest_create': (C.c_int, [C.POINTER(IngestConfig), c_vpp]),
    'hhx_ingest_push': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'hhx_ingest_push64': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'hhx_ingest_finalize': (C.c_int, [C.c_void_p, c_i64p, c_i64p]),
    'hhx_ingest_fetch': (C.c_int, [C.c_void_p] + [C.c_void_p] * 8),
    'hhx_ingest_flank_device': (C.c_int, [C.c_void_p, c_vpp, c_vpp, c_vpp]),
    'hhx_ingest_flank_count_device': (C.c_int, [C.c_void_p, c_vpp]),
    'hhx_ingest_destroy': (C.c_int, [C.c_void_p]),
    'hhx_ingest_keep_pairs': (C.c_int, [C.c_void_p, C.c_int]),
    'hhx_csr_vstack': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), c_vpp]),
    'hhx_mem_info': (C.c_int, [c_i64p, c_i64p]),
    'hhx_pool_cached_bytes': (C.c_int, [c_i64p]),
    'hhx_csr_pack_block': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'hhx_csr_unpack_blocks': (C.c_int, [C.c_int32, c_i64p, c_i64p, C.c_void_p, C.c_int64, C.c_int32, c_vpp]),
    'hhx_inflate_prune_keep': (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_vpp]),
    'hhx_mcl_resume': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, c_vpp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    'hhx_expand_links_dense': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_int, c_vpp, c_i64p, c_i64p]),
    'hhx_links_integer_ok': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'hhx_links_plan': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'hhx_dense_device': (C.c_int, [C.c_void_p, c_vpp, c_i64p, c_i32p, c_i32p]),
    'hhx_dense_inflate_prune': (C.c_int, [C.c_void_p, C.c_double, C.c_double, c_vpp]),
    'hhx_copy_rect_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    'hhx_transpose_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    'hhx_dense_inflate_prune_multi': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, c_vpp]),
    'hhx_dense_shape': (C.c_int, [C.c_void_p, c_i32p, c_i32p, c_i64p]),
    'hhx_dense_free': (C.c_int, [C.c_void_p]),
    'hhx_shard_create': (C.c_int, [C.c_void_p, C.c_void_p, c_vpp]),
    'hhx_shard_first': (C.c_int, [C.c_void_p, c_vpp]),
    'hhx_rank_first': (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, c_i32p]),
    'hhx_shard_emit': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_i32p, c_vpp, c_vpp, c_i64p]),
    'hhx_rows_from_entries': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, c_vpp]),
    'hhx_rows_from_runs': (C.c_int, [C.c_int32, c_i64p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int, c_vpp]),
    'hhx_shard_destroy': (C.c_int, [C.c_void_p]),
    'hhx_pairs_parser_create': (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    'hhx_pairs_parse': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, c_i64p, c_i64p]),
    'hhx_pairs_parser_arrays': (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5),
    'hhx_pairs_parser_fetch': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    'hhx_pairs_parser_set_wide': (C.c_int, [C.c_void_p, C.c_int]),
    'hhx_pairs_parser_fetch64': (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    'hhx_pairs_parser_bed_host': (C.c_int, [C.c_void_p, c_vpp, c_i64p]),
    'hhx_pairs_parser_destroy': (C.c_int, [C.c_void_p]),
    'hhx_pairs_format': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, c_i64p]),
    'hhx_bam_open': (C.c_int, [C.c_char_p, C.c_int, c_vpp]),
    'hhx_bam_header': (C.c_int, [C.c_void_p, c_i32p, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'hhx_bam_next': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int64, c_i64p, c_vpp, c_vpp, c_vpp, c_vpp]),
    'hhx_bam_fetch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_bam_close': (C.c_int, [C.c_void_p]),
    'hhx_contact_map_create': (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, c_vpp]),
    'hhx_contact_map_push': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32, c_i64p]),
    'hhx_contact_map_fetch': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hhx_contact_map_device': (C.c_int, [C.c_void_p, c_vpp, c_i32p]),
    'hhx_contact_map_destroy': (C.c_int, [C.c_void_p]),
    'hhx_ingest_fetch_ht_order': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hhx_ingest_keep_frag_pairs': (C.c_int, [C.c_void_p, C.c_int]),
    'hhx_ingest_fetch_frag_pairs': (C.c_int, [C.c_void_p, c_i64p, C.c_void_p, C.c_void_p]),
    'hhx_ingest_fetch_pairs': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_ingest_set_ordinal_base': (C.c_int, [C.c_void_p, C.c_int64]),
    'hhx_ingest_link_matrix': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_void_p, c_i32p, c_vpp]),
    'hhx_ingest_table_device': (C.c_int, [C.c_void_p, C.c_int, c_i64p, c_vpp, c_vpp, c_vpp, c_vpp, c_vpp]),
    'hhx_ingest_push_table': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_ingest_fetch_flank_values': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hhx_ingest_fetch_ht_items': (C.c_int, [C.c_void_p, c_i64p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'hhx_ingest_write_clm': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, c_i64p, c_i64p]),
    'hhx_write_link_pickle': (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, c_i64p]),
    'hhx_ingest_write_clm_async': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]),
    'hhx_ingest_write_link_pickle_async': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'hhx_write_link_pickle_async': (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'hhx_byte_sink_open': (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, c_vpp]),
    'hhx_byte_sink_reserve': (C.c_int, [C.c_void_p, C.c_int64, c_vpp]),
    'hhx_byte_sink_commit': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'hhx_byte_sink_close': (C.c_int, [C.c_void_p, c_i64p]),
    'hhx_pairs_parser_set_bed_sink': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hhx_text_reader_open': (C.c_int, [C.c_char_p, C.c_int64, C.c_int, c_vpp]),
    'hhx_text_reader_open_bgzf': (C.c_int, [C.c_char_p, C.c_int64, C.c_int, c_vpp]),
    'hhx_text_reader_next': (C.c_int, [C.c_void_p, c_vpp, c_i64p]),
    'hhx_text_reader_close': (C.c_int, [C.c_void_p]),
    'hhx_files_pending': (C.c_int, [c_i64p, c_i64p]),
    'hhx_files_join': (C.c_int, [c_i64p]),
}

_lib = None


def load():
    """dlopen the library and bind every declared symbol (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError('haphic_amd: %s is missing — build it with `python -m haphic_amd.build` '
                              '(hipcc --offload-arch=gfx950); there is no CPU fallback' % SO_PATH)
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1 and dlopens them by path; two HIP
        # runtimes in one process cannot both own the GPU.  Importing torch first makes the dynamic
        # linker resolve this library's DT_NEEDED libamdhip64.so.7 to the copy torch already loaded,
        # so torch tensors (device memory, streams, RCCL) and these kernels share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('libhaphic_hip: ' + load().hhx_last_error().decode('utf-8', 'replace'))


def device_count():
    n = C.c_int(0)
    check(load().hhx_device_count(C.byref(n)))
    return n.value


def tune(name, value):
    """which kernel class / arithmetic / layout the calls take (hhx_tune; same results under every setting); None: back to the default"""
    check(load().hhx_tune(name.encode(), -2 ** 63 if value is None else int(value)))


def profile_enable(on=True):
    check(load().hhx_profile_enable(int(on)))


def profile_reset():
    check(load().hhx_profile_reset())


def profile_get(kernel):
    ms, n = C.c_double(0), C.c_int64(0)
    check(load().hhx_profile_get(kernel.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value


def profile_counter(name):
    v = C.c_int64(0)
    check(load().hhx_profile_counter(name.encode(), C.byref(v)))
    return v.value


def ptr(a):
    """host pointer of a C-contiguous numpy array (or None)"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceCSR:
    """Device-resident CSR(T) == the reference's CSC(M) triple (hhx_csr handle)."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle

    @classmethod
    def from_arrays(cls, indptr, indices, data, n_cols=None):
        indptr, indices = np.asarray(indptr), np.asarray(indices)
        if len(indptr) and int(indptr[-1]) > np.iinfo(np.int32).max:       # a scipy matrix with int64 indices: do not wrap
            raise ValueError('matrix with {} entries exceeds the int32 index range of hhx_csr'.format(int(indptr[-1])))
        if indices.size and (int(indices.max()) > np.iinfo(np.int32).max or int(indices.min()) < 0):
            raise ValueError('column index outside the int32 range')
        indptr = np.ascontiguousarray(indptr, np.int32)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float32)
        n_rows = len(indptr) - 1
        out = C.c_void_p()
        check(load().hhx_csr_from_host(n_rows, n_rows if n_cols is None else n_cols, ptr(indptr), ptr(indices),
                                       ptr(data), C.byref(out)))
        return cls(out)

    @classmethod
    def from_scipy_csc(cls, m):
        """scipy CSC of M, canonicalised (sorted indices, duplicates summed) like the reference's own
        intermediate matrices after `.power()` (scipy _deduped_data)."""
        m = m.tocsc()
        if not m.has_canonical_format:
            m = m.copy()
            m.sum_duplicates()
        return cls.from_arrays(m.indptr, m.indices, m.data.astype(np.float32, copy=False), n_cols=m.shape[0])

    @classmethod
    def from_device(cls, n_rows, n_cols, nnz, indptr_ptr, indices_ptr, data_ptr):
        out = C.c_void_p()
        check(load().hhx_csr_from_device(n_rows, n_cols, nnz, C.c_void_p(indptr_ptr), C.c_void_p(indices_ptr),
                                         C.c_void_p(data_ptr), C.byref(out)))
        return cls(out)

    @property
    def shape3(self):
        r, c, z = C.c_int32(), C.c_int32(), C.c_int64()
        check(load().hhx_csr_shape(self.h, C.byref(r), C.byref(c), C.byref(z)))
        return r.value, c.value, z.value

    @property
    def nnz(self):
        return self.shape3[2]

    def to_arrays(self):
        r, c, z = self.shape3
        indptr = np.empty(r + 1, np.int32)
        indices = np.empty(z, np.int32)
        data = np.empty(z, np.float32)
        check(load().hhx_csr_to_host(self.h, ptr(indptr), ptr(indices), ptr(data)))
        return indptr, indices, data

    def to_scipy_csc(self):
        import scipy.sparse as sp
        r, c, _ = self.shape3
        indptr, indices, data = self.to_arrays()
        return sp.csc_matrix((data, indices, indptr), shape=(c, r))

    def device_ptrs(self):
        a, b, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(load().hhx_csr_device_ptrs(self.h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value or 0, b.value or 0, d.value or 0

    def copy(self):
        out = C.c_void_p()
        check(load().hhx_csr_copy(self.h, C.byref(out)))
        return DeviceCSR(out)

    def row_block(self, r0, r1):
        out = C.c_void_p()
        check(load().hhx_csr_row_block(self.h, r0, r1, C.byref(out)))
        return DeviceCSR(out)

    def free(self):
        if self.h is not None and self.h.value:
            load().hhx_csr_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---------------------------------------------------------------- thin functional wrappers
def normalize_l1(m):
    check(load().hhx_normalize_l1(m.h))
    return m


def inflate(m, inflation):
    check(load().hhx_inflate(m.h, float(inflation)))
    return m


def prune(m, pruning):
    out = C.c_void_p()
    check(load().hhx_prune(m.h, float(pruning), C.byref(out)))
    return DeviceCSR(out)


def inflate_prune(c, inflation, pruning):
    out = C.c_void_p()
    check(load().hhx_inflate_prune(c.h, float(inflation), float(pruning), C.byref(out)))
    return DeviceCSR(out)


def inflate_prune_keep(c, inflation, pruning):
    out = C.c_void_p()
    check(load().hhx_inflate_prune_keep(c.h, float(inflation), float(pruning), C.byref(out)))
    return DeviceCSR(out)


def vstack(blocks):
    arr = (C.c_void_p * len(blocks))(*[b.h.value for b in blocks])
    out = C.c_void_p()
    check(load().hhx_csr_vstack(len(blocks), arr, C.byref(out)))
    return DeviceCSR(out)


def mem_info():
    f, t = C.c_int64(0), C.c_int64(0)
    check(load().hhx_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def pool_prewarm(sizes):
    """hhx_pool_prewarm: blocks of these byte sizes from the driver into the pool's cache (call it from a helper thread while a long kernel runs)"""
    a = np.ascontiguousarray(sizes, np.int64)
    check(load().hhx_pool_prewarm(len(a), a.ctypes.data_as(c_i64p)))


def pool_cached_bytes():
    b = C.c_int64(0)
    check(load().hhx_pool_cached_bytes(C.byref(b)))
    return b.value


class DenseRows:
    """hhx_dense: rows [r0, r1) of the pre-expanded matrix M^2 as float32 in HBM — one expansion for a whole inflation sweep"""

    def __init__(self, links, r0, r1, fx_shift=52, upper_only=False):
        """upper_only: only the blocks (I, J >= I) of the rows are filled (integer arithmetic); the caller mirrors the rest"""
        self.h = C.c_void_p()
        f, z = C.c_int64(0), C.c_int64(0)
        check(load().hhx_expand_links_dense(links.h, int(r0), int(r1), int(fx_shift), int(bool(upper_only)), C.byref(self.h), C.byref(f), C.byref(z)))
        self.n_products, self.nnz_expanded = f.value, z.value
        self.n_rows, self.n_cols = int(r1) - int(r0), links.shape3[1]

    def device(self):
        """(device pointer of the block — n_rows rows of n_cols float32 —, row pitch in floats, columns per window, number of windows)"""
        x, ld, cap, nw = C.c_void_p(), C.c_int64(0), C.c_int32(0), C.c_int32(0)
        check(load().hhx_dense_device(self.h, C.byref(x), C.byref(ld), C.byref(cap), C.byref(nw)))
        return x.value or 0, ld.value, cap.value, nw.value

    def inflate_prune(self, inflation, pruning):
        """iteration 0 of mcl() (:2037-2042) of these rows at `inflation`"""
        out = C.c_void_p()
        check(load().hhx_dense_inflate_prune(self.h, float(inflation), float(pruning), C.byref(out)))
        return DeviceCSR(out)

    MULTI = 8       # inflations per pass of inflate_prune_multi (hhx_expand.hip: MULTI_MAX)

    def inflate_prune_multi(self, inflations, pruning):
        """iteration 0 at several inflations in one pass over the block per group of MULTI (hhx_dense_inflate_prune_multi): the
        matrices inflate_prune would return, in the order of `inflations`"""
        res = []
        inflations = [float(x) for x in inflations]
        for lo in range(0, len(inflations), self.MULTI):
            grp = inflations[lo:lo + self.MULTI]
            arr = (C.c_double * len(grp))(*grp)
            outs = (C.c_void_p * len(grp))()
            check(load().hhx_dense_inflate_prune_multi(self.h, len(grp), arr, float(pruning), outs))
            res.extend(DeviceCSR(o) for o in outs)
        return res

    def free(self):
        if self.h is not None and self.h.value:
            load().hhx_dense_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def links_integer_ok(links):
    """does iteration 0 on this raw link matrix run in the integer arithmetic (symmetric integer counts, row sums < 2^18)?"""
    ok, shift = C.c_int(0), C.c_int(0)
    check(load().hhx_links_integer_ok(links.h, C.byref(ok), C.byref(shift)))
    return bool(ok.value)


def links_plan(links):
    """(integer arithmetic?, layout of iteration 0: 0 all products into the fused epilogue / 1 square dense block / 2 upper block triangle)"""
    a, b = C.c_int(0), C.c_int(0)
    check(load().hhx_links_plan(links.h, C.byref(a), C.byref(b)))
    return bool(a.value), b.value


def mcl_resume(m, done, expansion, inflation, max_iter, pruning, want_stats=False):
    out = C.c_void_p()
    n_iter, conv = C.c_int(0), C.c_int(0)
    stats = np.zeros((max(int(max_iter), 1), 4), np.int64)
    check(load().hhx_mcl_resume(m.h, int(done), int(expansion), float(inflation), int(max_iter), float(pruning),
                                C.byref(out), C.byref(n_iter), C.byref(conv), ptr(stats)))
    res = (DeviceCSR(out), n_iter.value, bool(conv.value))
    return res + (stats[int(done):n_iter.value],) if want_stats else res


def spgemm(a, b, fx_shift=-1, want_products=False):
    out = C.c_void_p()
    f = C.c_int64(0)
    check(load().hhx_spgemm_ex(a.h, b.h, int(fx_shift), C.byref(out), C.byref(f) if want_products else None))
    return (DeviceCSR(out), f.value) if want_products else DeviceCSR(out)


def convergence_stat(m, last):
    s = C.c_float(0)
    check(load().hhx_convergence_stat(m.h, last.h, C.byref(s)))
    return s.value


def expand_inflate_prune(a, b, inflation, pruning, fx_shift=52):
    """one fused iteration: prune(normalize(power(a*b, r))); returns (matrix, n_products, nnz_expanded)"""
    out = C.c_void_p()
    f, z = C.c_int64(0), C.c_int64(0)
    check(load().hhx_expand_inflate_prune(a.h, b.h, int(fx_shift), float(inflation), float(pruning), C.byref(out),
                                          C.byref(f), C.byref(z)))
    return DeviceCSR(out), f.value, z.value


def link_weights(frag_i, frag_j, value, mode, n_frag, per_frag=None, tag=None, param=0.0, device_ptrs=None):
    """hhx_link_weights on host arrays (value: float64, rewritten in place) or, with device_ptrs = (i, j, value) device
    addresses of n = len(...) keys, on the device arrays of hhx_ingest_flank_device.  Returns the number of zeroed entries."""
    nz = C.c_int64(0)
    per = None if per_frag is None else np.ascontiguousarray(per_frag, np.int64)
    tg = None if tag is None else np.ascontiguousarray(tag, np.int32)
    if device_ptrs is not None:
        n, pi, pj, pv = device_ptrs
        check(load().hhx_link_weights(int(n), C.c_void_p(pi), C.c_void_p(pj), C.c_void_p(pv), 1, int(mode), int(n_frag), ptr(per), ptr(tg),
                                      float(param), C.byref(nz)))
        return nz.value
    assert value.dtype == np.float64 and value.flags.c_contiguous
    fi, fj = np.ascontiguousarray(frag_i, np.int32), np.ascontiguousarray(frag_j, np.int32)
    check(load().hhx_link_weights(len(value), ptr(fi), ptr(fj), ptr(value), 0, int(mode), int(n_frag), ptr(per), ptr(tg), float(param),
                                  C.byref(nz)))
    return nz.value


def names_blob(names):
    """names -> (uint8 array of the UTF-8 bytes back to back, int64 offsets [len(names) + 1])"""
    enc = [n.encode() for n in names]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum(np.fromiter(map(len, enc), np.int64, len(enc)), out=off[1:])
    blob = np.frombuffer(b''.join(enc) or b'\0', np.uint8)
    return blob, off


def write_link_pickle(path, i, j, count, names):
    """full_links.pkl / HT_links.pkl (output_pickle :710-715) straight from the link arrays: a protocol-4 pickle of
    `defaultdict(int)` {(names[i[k]], names[j[k]]): count[k]} in array order, written by the library's host code
    (hhx_write_link_pickle) without a Python object per key.  Returns the bytes written."""
    fi, fj = np.ascontiguousarray(i, np.int32), np.ascontiguousarray(j, np.int32)
    cnt = np.ascontiguousarray(count, np.int64)
    blob, off = names_blob(names)
    n_bytes = C.c_int64(0)
    check(load().hhx_write_link_pickle(os.fsencode(path), fi.size, ptr(fi), ptr(fj), ptr(cnt), len(names), ptr(blob), ptr(off),
                                       C.byref(n_bytes)))
    return n_bytes.value


# ---- the file-writer thread of the library (hhx_jobs.hip): output_pickle / output_clm return at once, the files are complete after files_join()
_pending_arrays = []                 # arrays of hhx_write_link_pickle_async that the library reads until the join
_join_registered = False


def files_async():
    """False: HAPHIC_SYNC_FILES=1 in the environment — the writers run on the caller's thread as they did before round 6"""
    return os.environ.get('HAPHIC_SYNC_FILES', '') not in ('1', 'true', 'yes')


def _register_join():
    global _join_registered
    if not _join_registered:
        import atexit
        atexit.register(_join_at_exit)
        _join_registered = True


def _join_at_exit():
    try:
        files_join()
    except RuntimeError as e:        # nobody is left to catch it: say it where a user sees it
        import sys
        print('haphic_amd: a file queued by output_pickle / output_clm was not written: %s' % e, file=sys.stderr)


def files_pending():
    """(files queued or being written, files finished since the library was loaded)"""
    a, b = C.c_int64(0), C.c_int64(0)
    check(load().hhx_files_pending(C.byref(a), C.byref(b)))
    return a.value, b.value


def files_join():
    """wait for every queued file (hhx_files_join); RuntimeError with the first failure if a writer failed"""
    if _lib is None:
        return
    n = C.c_int64(0)
    rc = _lib.hhx_files_join(C.byref(n))
    del _pending_arrays[:]
    check(rc)


class ByteSink:
    """hhx_byte_sink: a file fed from device buffers through the library's file-writer thread (alignments.bed, deferred)"""

    def __init__(self, path, hbm_budget_bytes=0, expected_bytes=0):
        self.h = C.c_void_p()
        _register_join()
        check(load().hhx_byte_sink_open(os.fsencode(path), int(hbm_budget_bytes), int(expected_bytes), C.byref(self.h)))

    def close(self):
        """queue the close (the file is complete after files_join()); returns the bytes handed to the sink"""
        n = C.c_int64(0)
        if self.h is not None and self.h.value:
            check(load().hhx_byte_sink_close(self.h, C.byref(n)))
            self.h = None
        return n.value


class TextReader:
    """hhx_text_reader: a text file as chunks of whole lines in pinned host memory, read ahead by threads of the library"""

    def __init__(self, path, chunk_bytes=256 << 20, threads=4, bgzf=False):
        """bgzf: the file is bgzipped (BGZF blocks inflated by the threads); RuntimeError('... is not a BGZF file ...') for anything else"""
        self.h = C.c_void_p()
        fn = load().hhx_text_reader_open_bgzf if bgzf else load().hhx_text_reader_open
        check(fn(os.fsencode(path), int(chunk_bytes), int(threads), C.byref(self.h)))

    def __iter__(self):
        """(host pointer, bytes) per chunk; a pointer is valid until the next one is asked for"""
        host, n = C.c_void_p(), C.c_int64(0)
        while True:
            check(load().hhx_text_reader_next(self.h, C.byref(host), C.byref(n)))
            if n.value == 0:
                return
            yield host.value, n.value

    def close(self):
        if self.h is not None and self.h.value:
            load().hhx_text_reader_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_link_pickle_async(path, i, j, count, names):
    """write_link_pickle on the library's file-writer thread; the arrays are kept alive here until files_join()"""
    fi, fj = np.ascontiguousarray(i, np.int32), np.ascontiguousarray(j, np.int32)
    cnt = np.ascontiguousarray(count, np.int64)
    blob, off = names_blob(names)
    _register_join()
    _pending_arrays.append((fi, fj, cnt))
    check(load().hhx_write_link_pickle_async(os.fsencode(path), fi.size, ptr(fi), ptr(fj), ptr(cnt), len(names), ptr(blob), ptr(off)))


def group_link_sums(frag_i, frag_j, links, group, n_groups):
    """hhx_group_link_sums: (sums, first) int64 [n_ctg, n_groups]; first == -1 where a cell got no contribution"""
    fi, fj = np.ascontiguousarray(frag_i, np.int32), np.ascontiguousarray(frag_j, np.int32)
    lk = np.ascontiguousarray(links, np.int64)
    grp = np.ascontiguousarray(group, np.int32)
    sums = np.zeros((len(grp), int(n_groups)), np.int64)
    first = np.full((len(grp), int(n_groups)), -1, np.int64)
    check(load().hhx_group_link_sums(len(lk), ptr(fi), ptr(fj), ptr(lk), len(grp), ptr(grp), int(n_groups), ptr(sums), ptr(first)))
    return sums, first


def row_products(a, b):
    """products per row of a * b (int64 numpy)"""
    out = np.zeros(max(a.shape3[0], 1), np.int64)
    check(load().hhx_row_products(a.h, b.h, ptr(out)))
    return out[:a.shape3[0]]


def expand_links(links, r0, r1, inflation, pruning, fx_shift=52):
    """iteration 0 of mcl() for the rows [r0, r1) of the RAW link matrix `links` (the whole matrix: the right operand)"""
    out = C.c_void_p()
    f, z = C.c_int64(0), C.c_int64(0)
    check(load().hhx_expand_links(links.h, int(r0), int(r1), int(fx_shift), float(inflation), float(pruning), C.byref(out), C.byref(f), C.byref(z)))
    return DeviceCSR(out), f.value, z.value


def mcl(pre_expanded, expansion, inflation, max_iter, pruning, want_stats=False, normalized=False, links=False):
    """normalized=False: the reference seam (matrix already pre-expanded); True: start from the
    L1-normalised link matrix, pre-expansion fused into iteration 0 (hhx_mcl_normalized); links=True: start
    from the raw link matrix of dict_to_matrix (hhx_mcl_links: normalisation fused too)."""
    out = C.c_void_p()
    n_iter, conv = C.c_int(0), C.c_int(0)
    stats = np.zeros((max(int(max_iter), 1), 4), np.int64)
    fn = load().hhx_mcl_links if links else (load().hhx_mcl_normalized if normalized else load().hhx_mcl)
    check(fn(pre_expanded.h, int(expansion), float(inflation), int(max_iter), float(pruning),
                         C.byref(out), C.byref(n_iter), C.byref(conv), ptr(stats)))
    res = (DeviceCSR(out), n_iter.value, bool(conv.value))
    return res + (stats[:n_iter.value],) if want_stats else res


def interpret(m):
    r, _, z = m.shape3
    att = np.empty(max(r, 1), np.int32)
    att_ptr = np.empty(r + 1, np.int32)
    members = np.empty(max(z, 1), np.int32)
    na = C.c_int32(0)
    check(load().hhx_interpret(m.h, ptr(att), ptr(att_ptr), ptr(members), C.byref(na)))
    na = na.value
    return att[:na], att_ptr[:na + 1], members[:att_ptr[na]]


def dict_to_matrix(frag_i, frag_j, value, n_frag, in_set, n_rest, add_self_loops=True, on_device=False, n_keys=None):
    """frag_i/frag_j/value: numpy arrays (on_device=False) or raw device pointers (ints, on_device=True)."""
    in_set = np.ascontiguousarray(in_set, np.uint8)
    frag_index = np.empty(max(n_frag, 1), np.int32)
    n_linked = C.c_int32(0)
    out = C.c_void_p()
    if on_device:
        a, b, v = C.c_void_p(frag_i), C.c_void_p(frag_j), C.c_void_p(value)
        nk = int(n_keys)
    else:
        fi = np.ascontiguousarray(frag_i, np.int32)
        fj = np.ascontiguousarray(frag_j, np.int32)
        fv = np.ascontiguousarray(value, np.float64)
        a, b, v = ptr(fi), ptr(fj), ptr(fv)
        nk = fi.size
    check(load().hhx_dict_to_matrix(nk, a, b, v, int(on_device), int(n_frag), ptr(in_set), int(n_rest),
                                    int(add_self_loops), ptr(frag_index), C.byref(n_linked), C.byref(out)))
    return DeviceCSR(out), frag_index[:n_frag], n_linked.value


def rank_sums(m, topN):
    """filter_fragments' rank-sum statistic of every row of the (self-loop free) link matrix"""
    out = np.zeros(max(m.shape3[0], 1), np.int64)
    check(load().hhx_rank_sums(m.h, int(topN), ptr(out)))
    return out[:m.shape3[0]]


def count_re_sites(seq, seg_off, seg_len, sites):
    """seq: bytes-like / uint8 array; sites: list of bytes patterns (already N-expanded).  Returns int64 counts."""
    buf = np.frombuffer(seq, np.uint8) if not isinstance(seq, np.ndarray) else np.ascontiguousarray(seq, np.uint8)
    off = np.ascontiguousarray(seg_off, np.int64)
    ln = np.ascontiguousarray(seg_len, np.int64)
    pats = np.frombuffer(b''.join(sites), np.uint8) if sites else np.zeros(1, np.uint8)
    plen = np.array([len(x) for x in sites], np.int32) if sites else np.zeros(1, np.int32)
    out = np.zeros(max(off.size, 1), np.int64)
    check(load().hhx_count_re_sites(ptr(buf) if buf.size else None, buf.size, off.size, ptr(off), ptr(ln), len(sites), ptr(pats),
                                    ptr(plen), ptr(out)))
    return out[:off.size]


class BamReader:
    """hhx_bam: BGZF / BAM container -> batches of device id / position arrays (f4)."""

    def __init__(self, path, threads=0):
        self.h = C.c_void_p()
        check(load().hhx_bam_open(os.fsencode(path), int(threads), C.byref(self.h)))
        n_ref, text, tlen = C.c_int32(0), C.c_char_p(), C.c_int64(0)
        names, offs = C.c_void_p(), C.c_void_p()
        check(load().hhx_bam_header(self.h, C.byref(n_ref), C.byref(text), C.byref(tlen), C.byref(names), C.byref(offs)))
        self.header_text = C.string_at(text, tlen.value).decode('utf-8', 'replace') if tlen.value else ''
        off = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_int64)), (n_ref.value + 1,)).copy() if n_ref.value else np.zeros(1, np.int64)
        blob = C.string_at(names, int(off[-1])) if n_ref.value and off[-1] else b''
        self.ref_names = [blob[off[k]:off[k + 1]].decode() for k in range(n_ref.value)]
        self._map = None

    def set_contigs(self, ctg_ids):
        """ctg_ids: name -> contig id of the FASTA; BAM references outside it map to -1 (`ref not in fa_dict`)"""
        self._map = np.fromiter((ctg_ids.get(n, -1) for n in self.ref_names), np.int32, len(self.ref_names))
        if not len(self._map):
            self._map = np.zeros(0, np.int32)

    def next_batch(self, need_flags, drop_same_ref, max_inflated_bytes=256 << 20):
        """-> (n_records, [id1, pos1, id2, pos2] device pointers); n_records == 0 at the end of the file"""
        n = C.c_int64(0)
        p = [C.c_void_p() for _ in range(4)]
        check(load().hhx_bam_next(self.h, int(need_flags), int(drop_same_ref), len(self.ref_names), ptr(self._map) if len(self._map) else None,
                                  int(max_inflated_bytes), C.byref(n), *[C.byref(x) for x in p]))
        self.last_n = n.value
        return n.value, [x.value for x in p]

    def fetch(self):
        """host copies (id1, pos1, id2, pos2) of the last batch"""
        out = [np.empty(self.last_n, np.int32) for _ in range(4)]
        check(load().hhx_bam_fetch(self.h, *[ptr(a) for a in out]))
        return out

    def close(self):
        if self.h:
            load().hhx_bam_close(self.h)
            self.h = None


class ContactMap:
    """hhx_contact_map: the dense scaffold-bin contact matrix of `haphic plot` (f4).  Tables as include/haphic_hip.h lists them."""

    def __init__(self, in_set, aln_ptr, list_ptr, seg_lo, seg_hi, seg_bin, bin_size, n_total_bins):
        self.in_set = np.ascontiguousarray(in_set, np.uint8)
        self.aln_ptr = np.ascontiguousarray(aln_ptr, np.int64)
        self.list_ptr = np.ascontiguousarray(list_ptr, np.int32)
        self.seg = [np.ascontiguousarray(a, np.int32) for a in (seg_lo, seg_hi, seg_bin)]
        self.n_bins = int(n_total_bins)
        self.h = C.c_void_p()
        check(load().hhx_contact_map_create(len(self.in_set), ptr(self.in_set), ptr(self.aln_ptr), ptr(self.list_ptr), len(self.seg[0]),
                                            ptr(self.seg[0]), ptr(self.seg[1]), ptr(self.seg[2]), int(bin_size), self.n_bins, C.byref(self.h)))

    def _push(self, n, p, on_device, pos_offset):
        bad = C.c_int64(-1)
        check(load().hhx_contact_map_push(self.h, int(n), p[0], p[1], p[2], p[3], int(on_device), int(pos_offset), C.byref(bad)))
        return bad.value

    def push(self, id1, pos1, id2, pos2, pos_offset=0):
        """host int32 arrays; -> -1, or 2 * k + side of the first pair whose position is outside the AGP"""
        arrs = [np.ascontiguousarray(a, np.int32) for a in (id1, pos1, id2, pos2)]
        return self._push(len(arrs[0]), [ptr(a) for a in arrs], False, pos_offset)

    def push_device(self, n, id1, pos1, id2, pos2, pos_offset=0):
        return self._push(n, [C.c_void_p(x) for x in (id1, pos1, id2, pos2)], True, pos_offset)

    def fetch(self):
        out = np.empty((self.n_bins, self.n_bins), np.int64)
        check(load().hhx_contact_map_fetch(self.h, ptr(out)))
        return out

    def destroy(self):
        if self.h:
            load().hhx_contact_map_destroy(self.h)
            self.h = None


class PairsParser:
    """hhx_pairs_parser: .pairs text chunks -> device id / position arrays (+ the alignments.bed bytes)."""

    def __init__(self, names):
        enc = [n.encode() for n in names]
        off = np.zeros(len(enc) + 1, np.int64)
        if enc:
            off[1:] = np.cumsum([len(b) for b in enc])
        blob = np.frombuffer(b''.join(enc) or b'\0', np.uint8)
        self.h = C.c_void_p()
        check(load().hhx_pairs_parser_create(len(enc), ptr(blob), ptr(off), C.byref(self.h)))
        self.n_lines = self.bed_bytes = 0
        self.wide = False

    def set_wide(self, on=True):
        """positions as int64 from the next parse on (contigs of 2^31 bp and more)"""
        check(load().hhx_pairs_parser_set_wide(self.h, int(on)))
        self.wide = bool(on)

    def parse(self, text, want_bed=False, device_ptr=None, n_bytes=None, host_ptr=None):
        """text: bytes-like holding whole lines (or device_ptr / host_ptr + n_bytes); raises IndexError / ValueError like
        the reference's cols[k] / int() do"""
        nl, nb = C.c_int64(0), C.c_int64(0)
        if host_ptr is not None:
            rc = load().hhx_pairs_parse(self.h, C.c_void_p(host_ptr), int(n_bytes), 0, int(want_bed), C.byref(nl), C.byref(nb))
        elif device_ptr is None:
            buf = np.frombuffer(text, np.uint8)
            rc = load().hhx_pairs_parse(self.h, ptr(buf) if buf.size else None, buf.size, 0, int(want_bed), C.byref(nl), C.byref(nb))
        else:
            rc = load().hhx_pairs_parse(self.h, C.c_void_p(device_ptr), int(n_bytes), 1, int(want_bed), C.byref(nl), C.byref(nb))
        if rc:
            msg = load().hhx_last_error().decode('utf-8', 'replace')
            kind = {'IndexError': IndexError, 'ValueError': ValueError}.get(msg.split(':', 1)[0])
            if kind:
                raise kind(msg.split(': ', 1)[1])
            raise RuntimeError('libhaphic_hip: ' + msg)
        self.n_lines, self.bed_bytes = nl.value, nb.value
        return self.n_lines

    def device_arrays(self):
        p = [C.c_void_p() for _ in range(5)]
        check(load().hhx_pairs_parser_arrays(self.h, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def fetch(self, want_bed=False):
        out = [np.empty(self.n_lines, np.int64 if self.wide and k in (1, 3) else np.int32) for k in range(4)]
        bed = np.empty(self.bed_bytes if want_bed else 0, np.uint8)
        fn = load().hhx_pairs_parser_fetch64 if self.wide else load().hhx_pairs_parser_fetch
        check(fn(self.h, *[ptr(a) for a in out], ptr(bed) if bed.size else None))
        return out + [bed.tobytes()]

    def format_pairs(self, n, id1_ptr, pos1_ptr, id2_ptr, pos2_ptr, first_read=0, text_ptr=None, capacity=0):
        """measurement only: device id / position arrays -> .pairs text on the device (hhx_pairs_format); returns the byte count
        (text_ptr None: just the size)"""
        nb = C.c_int64(0)
        check(load().hhx_pairs_format(self.h, int(n), C.c_void_p(id1_ptr), C.c_void_p(pos1_ptr), C.c_void_p(id2_ptr), C.c_void_p(pos2_ptr), int(first_read),
                                      C.c_void_p(text_ptr) if text_ptr else None, int(capacity), C.byref(nb)))
        return nb.value

    def set_bed_sink(self, sink):
        """from the next parse(want_bed=True) on the BED records are formatted into the ByteSink's ring in HBM and queued for the file-writer
        thread (None: back to the parser's own buffer)"""
        check(load().hhx_pairs_parser_set_bed_sink(self.h, sink.h if sink is not None else None))

    def bed_host(self):
        """the alignments.bed bytes of the last parse as a uint8 VIEW of pinned memory owned by the parser (valid until the
        second following call)"""
        host, n = C.c_void_p(), C.c_int64(0)
        check(load().hhx_pairs_parser_bed_host(self.h, C.byref(host), C.byref(n)))
        if not n.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), (n.value,))

    def fetch_bed(self):
        """the alignments.bed bytes of the last parse (uint8 array)"""
        bed = np.empty(self.bed_bytes, np.uint8)
        if self.bed_bytes:
            check(load().hhx_pairs_parser_fetch(self.h, None, None, None, None, ptr(bed)))
        return bed

    def destroy(self):
        if self.h is not None and self.h.value:
            load().hhx_pairs_parser_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Ingest:
    """hhx_ingest handle: push batches of (id1, pos1, id2, pos2), then finalize/fetch."""

    def __init__(self, table, flank, bins=False, skip_intra=False, expected_keys=0):
        self._keep = table      # keeps the numpy arrays alive during create
        cfg = IngestConfig()
        cfg.n_ctg, cfg.n_frag = table.n_ctg, table.n_frag
        cfg.ctg_rank = table.ctg_rank.ctypes.data_as(c_i32p)
        cfg.ctg_len = table.ctg_len.ctypes.data_as(c_i64p)
        cfg.ctg_frag0 = table.ctg_frag0.ctypes.data_as(c_i32p)
        cfg.ctg_split = table.ctg_split.ctypes.data_as(c_u8p)
        cfg.frag_rank = table.frag_rank.ctypes.data_as(c_i32p)
        cfg.frag_len = table.frag_len.ctypes.data_as(c_i64p)
        cfg.frag_nx = table.frag_nx.ctypes.data_as(c_u8p)
        cfg.bin_size, cfg.flank = int(table.bin_size), int(flank)
        cfg.bins, cfg.skip_intra = int(bool(bins)), int(bool(skip_intra))
        cfg.expected_keys = int(expected_keys)
        self.n_frag = table.n_frag
        self.h = C.c_void_p()
        check(load().hhx_ingest_create(C.byref(cfg), C.byref(self.h)))
        self.n_full = self.n_flank = None

    def push(self, id1, pos1, id2, pos2, wide=None):
        """host arrays.  wide=True: 64-bit positions through hhx_ingest_push64 (contigs of 2^31 bp and more, :116-147).  wide=None:
        decided by the VALUES, not the dtype — default-int64 numpy arrays whose positions fit 32 bits take the packed 32-bit path
        (half the host-to-device bytes, four pairs per lane in the map kernel)."""
        pos = [np.asarray(pos1), np.asarray(pos2)]
        i32max = np.iinfo(np.int32).max
        # by VALUE, whatever the dtype (a uint32 array can hold 2^31 and more; an int64 array usually does not)
        beyond = any(a.dtype != np.int32 and a.size and int(a.max()) > i32max for a in pos)
        if wide is None:
            wide = beyond
        elif not wide and beyond:
            raise ValueError('Ingest.push(wide=False): a position exceeds the int32 range; pass wide=True (hhx_ingest_push64)')
        if any(a.dtype != np.int32 and a.dtype.kind == 'i' and a.size and int(a.min()) < -i32max - 1 for a in pos):
            raise ValueError('Ingest.push: a position below the int32 range')
        if wide:
            ids = [np.ascontiguousarray(a, np.int32) for a in (id1, id2)]
            pos = [np.ascontiguousarray(a, np.int64) for a in (pos1, pos2)]
            check(load().hhx_ingest_push64(self.h, ids[0].size, ptr(ids[0]), ptr(pos[0]), ptr(ids[1]), ptr(pos[1]), 0))
            return
        arrs = [np.ascontiguousarray(a, np.int32) for a in (id1, pos1, id2, pos2)]
        check(load().hhx_ingest_push(self.h, arrs[0].size, *[ptr(a) for a in arrs], 0))

    def push_device(self, n_pairs, id1_ptr, pos1_ptr, id2_ptr, pos2_ptr, wide=False):
        """device arrays (int32 ids; positions int32, or int64 with wide=True)"""
        fn = load().hhx_ingest_push64 if wide else load().hhx_ingest_push
        check(fn(self.h, int(n_pairs), C.c_void_p(id1_ptr), C.c_void_p(pos1_ptr), C.c_void_p(id2_ptr), C.c_void_p(pos2_ptr), 1))

    def finalize(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(load().hhx_ingest_finalize(self.h, C.byref(a), C.byref(b)))
        self.n_full, self.n_flank = a.value, b.value
        return self.n_full, self.n_flank

    _FETCH = ('full_i', 'full_j', 'full_cnt', 'ht_cnt', 'flank_i', 'flank_j', 'flank_cnt', 'frag_links')

    def fetch(self, want=None):
        """host copies in dict insertion order (hhx_ingest_fetch); want: the subset of _FETCH to copy (default: all)"""
        if self.n_full is None:
            self.finalize()
        nf, nk = self.n_full, self.n_flank
        shape = dict(full_i=(nf, np.int32), full_j=(nf, np.int32), full_cnt=(nf, np.int64), ht_cnt=((nf, 4), np.int64),
                     flank_i=(nk, np.int32), flank_j=(nk, np.int32), flank_cnt=(nk, np.int64), frag_links=(self.n_frag, np.int64))
        want = self._FETCH if want is None else tuple(want)
        out = {k: np.empty(*shape[k]) for k in want}
        check(load().hhx_ingest_fetch(self.h, *[ptr(out[k]) if k in out else None for k in self._FETCH]))
        return out

    def fetch_flank_values(self):
        """the float64 values of the flank table in dict order (counts, or the weights hhx_link_weights left there)"""
        out = np.empty(self.n_flank, np.float64)
        check(load().hhx_ingest_fetch_flank_values(self.h, ptr(out)))
        return out

    def weigh_flank(self, mode, per_frag=None, tag=None, param=0.0):
        """hhx_link_weights on the device-resident flank table (normalize_by_nlinks :718-724 and friends): the weights never
        leave HBM on their way into link_matrix(weighted=True)"""
        pi, pj, pv = self.flank_device()
        return link_weights(None, None, None, mode, self.n_frag, per_frag=per_frag, tag=tag, param=param,
                            device_ptrs=(self.n_flank, pi, pj, pv))

    def flank_device(self):
        a, b, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(load().hhx_ingest_flank_device(self.h, C.byref(a), C.byref(b), C.byref(v)))
        return a.value or 0, b.value or 0, v.value or 0

    def flank_count_device(self):
        c = C.c_void_p()
        check(load().hhx_ingest_flank_count_device(self.h, C.byref(c)))
        return c.value or 0

    def keep_pairs(self, on=True):
        """also keep the oriented coordinates of every counted pair (CLM / coordinate side products)"""
        check(load().hhx_ingest_keep_pairs(self.h, int(on)))

    def fetch_pairs(self, max_read_pairs, full_cnt):
        """(clm_ptr, clm, crd_ptr, crd) in dict insertion order; clm_ptr counts READ PAIRS (x 4 values each)"""
        nf = len(full_cnt)
        total = int(np.sum(full_cnt))
        capped = int(np.minimum(full_cnt, max_read_pairs).sum()) if max_read_pairs > 0 else 0
        clm_ptr = np.zeros(nf + 1, np.int64)
        crd_ptr = np.zeros(nf + 1, np.int64)
        clm = np.zeros(max(4 * total, 1), np.int64)
        crd = np.zeros(max(2 * capped, 1), np.int64)
        check(load().hhx_ingest_fetch_pairs(self.h, int(max_read_pairs), ptr(clm_ptr), ptr(clm), ptr(crd_ptr), ptr(crd)))
        return clm_ptr, clm[:4 * total], crd_ptr, crd[:2 * capped]

    def fetch_ht_order(self):
        """[n_full, 4] stream positions of the first pair per head/tail quadrant (INT64_MAX: none): HT_link_dict's order"""
        first = np.full((max(self.n_full, 1), 4), np.iinfo(np.int64).max, np.int64)
        check(load().hhx_ingest_fetch_ht_order(self.h, ptr(first)))
        return first[:self.n_full]

    def fetch_ht_items(self):
        """HT_link_dict in insertion order: (name_i, name_j, count) with name ids 2 * contig + (1 for '_T'); ordered on the device"""
        n = C.c_int64(0)
        check(load().hhx_ingest_fetch_ht_items(self.h, C.byref(n), None, None, None))
        ni, nj, cnt = np.empty(n.value, np.int32), np.empty(n.value, np.int32), np.empty(n.value, np.int64)
        if n.value:
            check(load().hhx_ingest_fetch_ht_items(self.h, C.byref(n), ptr(ni), ptr(nj), ptr(cnt)))
        return ni, nj, cnt

    def n_ht_items(self):
        n = C.c_int64(0)
        check(load().hhx_ingest_fetch_ht_items(self.h, C.byref(n), None, None, None))
        return n.value

    def keep_frag_pairs(self, on=True):
        check(load().hhx_ingest_keep_frag_pairs(self.h, int(on)))

    def fetch_frag_pairs(self):
        """every distinct oriented fragment pair of the stream (ctg_pair_to_frag :1731): (frag_i, frag_j) arrays"""
        n = C.c_int64(0)
        check(load().hhx_ingest_fetch_frag_pairs(self.h, C.byref(n), None, None))
        fi, fj = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        if n.value:
            check(load().hhx_ingest_fetch_frag_pairs(self.h, C.byref(n), ptr(fi), ptr(fj)))
        return fi, fj

    def set_ordinal_base(self, base):
        """global stream ordinal of this handle's first pair (multi-GPU chunk offset); before the first push"""
        check(load().hhx_ingest_set_ordinal_base(self.h, int(base)))

    def link_matrix(self, in_set, n_rest=-1, add_self_loops=True, weighted=False):
        """dict_to_matrix fused onto the device-resident flank table: (DeviceCSR, frag_index, n_linked).  weighted: the values
        are the float64 weights of weigh_flank (hhx_dict_to_matrix over the ordered device arrays) instead of the counts."""
        if self.n_full is None:
            self.finalize()
        in_set = np.ascontiguousarray(in_set, np.uint8)
        if weighted:
            pi, pj, pv = self.flank_device()
            return dict_to_matrix(pi, pj, pv, self.n_frag, in_set, n_rest, add_self_loops=add_self_loops, on_device=True,
                                  n_keys=self.n_flank)
        frag_index = np.empty(max(self.n_frag, 1), np.int32)
        n_linked = C.c_int32(0)
        out = C.c_void_p()
        check(load().hhx_ingest_link_matrix(self.h, ptr(in_set), int(n_rest), int(add_self_loops), ptr(frag_index),
                                            C.byref(n_linked), C.byref(out)))
        return DeviceCSR(out), frag_index[:self.n_frag], n_linked.value

    def write_clm(self, path, ctg_names):
        """paired_links.clm (output_clm :376-392) from the kept read pairs: grouped, sorted and formatted on the device.
        Returns (lines, bytes) written."""
        blob, off = names_blob(ctg_names)
        n_lines, n_bytes = C.c_int64(0), C.c_int64(0)
        check(load().hhx_ingest_write_clm(self.h, os.fsencode(path), ptr(blob), ptr(off), C.byref(n_lines), C.byref(n_bytes)))
        return n_lines.value, n_bytes.value

    def write_clm_async(self, path, ctg_names, drop_pairs=False):
        """write_clm on the library's file-writer thread (hhx_ingest_write_clm_async): checked and opened here, complete after
        files_join(); drop_pairs: the kept read pairs leave HBM when the file is done"""
        blob, off = names_blob(ctg_names)
        _register_join()
        check(load().hhx_ingest_write_clm_async(self.h, os.fsencode(path), ptr(blob), ptr(off), int(bool(drop_pairs))))

    PICKLE_KINDS = {'full': 0, 'HT': 1, 'flank': 2}

    def write_link_pickle_async(self, kind, path, names):
        """full_links.pkl / HT_links.pkl (output_pickle :710-715) of this handle's table on the file-writer thread: the items are
        fetched (HT: ordered on the device) and encoded there; complete after files_join()"""
        blob, off = names_blob(names)
        _register_join()
        check(load().hhx_ingest_write_link_pickle_async(self.h, self.PICKLE_KINDS[kind], os.fsencode(path), len(names), ptr(blob), ptr(off)))

    def table_device(self, which=0):
        """aggregated table (unordered): (n_rows, key_ptr, ord_full_ptr, ord_flank_ptr, ht_ptr, flank_ptr)"""
        if self.n_full is None:
            self.finalize()
        n = C.c_int64(0)
        p = [C.c_void_p() for _ in range(5)]
        check(load().hhx_ingest_table_device(self.h, int(which), C.byref(n), *[C.byref(x) for x in p]))
        return (n.value,) + tuple(x.value or 0 for x in p)

    def push_table(self, which, n_rows, key_ptr, ord_full_ptr, ord_flank_ptr, ht_ptr, flank_ptr):
        check(load().hhx_ingest_push_table(self.h, int(which), int(n_rows), C.c_void_p(key_ptr), C.c_void_p(ord_full_ptr),
                                           C.c_void_p(ord_flank_ptr), C.c_void_p(ht_ptr), C.c_void_p(flank_ptr)))

    def destroy(self):
        if self.h is not None and self.h.value:
            load().hhx_ingest_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
