// dict_to_matrix, scripts/HapHiC_cluster.py:310-373, on the device: link table -> symmetric float32
// CSR(T) (== the reference's CSC) with unit self loops.
//
// Two front ends share the kernels:
//   hhx_dict_to_matrix        rows (frag_i, frag_j, value) ALREADY in dict insertion order (the S4 seam):
//                             the row position k is the key's ordinal;
//   hhx_link_matrix_from_run  an aggregated ingest table in arbitrary order whose rows carry the stream
//                             ordinal of their first flank-qualified pair (hhx_ingest_link_matrix): the
//                             insertion order is never materialised.
// Index assignment (:337-349): a fragment's matrix index is the rank of its first appearance when the
// items are scanned in insertion order, i before j — i.e. the rank of min over its keys of
// 2 * ordinal + side.  atomicMin per fragment, then the rank of that position among the fragments (tiled
// all-pairs count, no sort).  Rows are filled with atomic cursors and put in column order with an LDS
// bitmap rank (columns of a row are unique).
#include "hhx_ingest.h"
#include "hhx_partition.h"
#include "hhx_sort.h"

using namespace hhx;

int hhx_csr_alloc_internal(i32 n_rows, i32 n_cols, i64 nnz, hhx_csr **out);

namespace {

struct RowsView {           // insertion-ordered arrays
    const i32 *fi, *fj;
    const double *val;
    __device__ __forceinline__ bool get(i64 k, i32 &a, i32 &b, u64 &ord, float &v) const {
        a = fi[k]; b = fj[k]; ord = (u64)k;
        v = (float)val[k];                                      // dtype=float32 at :368
        return true;
    }
};
struct RunView {            // aggregated ingest table
    const u64 *key, *ord_flank;
    const u32 *fl;
    __device__ __forceinline__ bool get(i64 k, i32 &a, i32 &b, u64 &ord, float &v) const {
        ord = ord_flank[k];
        if (ord == NO_ORD) return false;                        // the key never entered flank_link_dict
        a = (i32)(key[k] >> ID_BITS); b = (i32)(key[k] & ID_MASK);
        v = (float)fl[k];
        return true;
    }
};

template <class View>
__global__ __launch_bounds__(256) void k_first_pos(View vw, i64 n_keys, const unsigned char *__restrict__ in_set,
                                                   unsigned long long *first_pos) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        i32 a, b; u64 ord; float v;
        if (!vw.get(k, a, b, ord, v) || !in_set[a] || !in_set[b]) continue;
        atomicMin(&first_pos[a], (unsigned long long)(2 * ord));
        atomicMin(&first_pos[b], (unsigned long long)(2 * ord + 1));
    }
}
// Matrix index of a fragment = number of linked fragments whose first position is smaller.  The positions
// 2 * ordinal + side are distinct (a position names one side of one read pair) but sparse in
// [0, 2 * #pairs), so they are ranked by a tiled all-pairs count over the n_frag values (LDS broadcast
// tiles; 10^10 compares at n = 100k, a fraction of a millisecond on 256 CUs) instead of a bitmap.
// `none` marks a fragment without entries: ~0 here, INT64_MAX in the multi-GPU build (its all-reduce(min) is signed)
__global__ __launch_bounds__(256) void k_rank_first(i32 n_frag, const unsigned long long *__restrict__ first_pos,
                                                    i32 *__restrict__ frag_index, unsigned int *n_linked, unsigned long long none = ~0ull) {
    __shared__ unsigned long long tile[1024];
    const i32 f = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long mine = f < n_frag ? first_pos[f] : none;
    i32 rank = 0;
    for (i32 t0 = 0; t0 < n_frag; t0 += 1024) {
        for (i32 t = threadIdx.x; t < 1024; t += blockDim.x) tile[t] = (t0 + t < n_frag) ? first_pos[t0 + t] : none;
        __syncthreads();
        if (mine != none) {
#pragma unroll 8
            for (i32 t = 0; t < 1024; ++t) rank += tile[t] < mine;
        }
        __syncthreads();
    }
    if (f < n_frag) {
        frag_index[f] = mine == none ? -1 : rank;
        if (mine != none) atomicAdd(n_linked, 1u);
    }
}
template <class View>
__global__ __launch_bounds__(256) void k_row_counts(View vw, i64 n_keys, const i32 *__restrict__ frag_index, i32 *cnt) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        i32 a, b; u64 ord; float v;
        if (!vw.get(k, a, b, ord, v)) continue;
        a = frag_index[a]; b = frag_index[b];
        if (a < 0 || b < 0) continue;
        atomicAdd(&cnt[a], 1);
        atomicAdd(&cnt[b], 1);
    }
}
__global__ __launch_bounds__(256) void k_init_counts(i32 shape, i32 *cnt, i32 v) {
    for (i32 r = blockIdx.x * blockDim.x + threadIdx.x; r < shape; r += gridDim.x * blockDim.x) cnt[r] = v;
}
// unsorted fill (atomic cursors) ...
template <class View>
__global__ __launch_bounds__(256) void k_fill(View vw, i64 n_keys, const i32 *__restrict__ frag_index, const i32 *__restrict__ indptr,
                                              i32 *cursor, i32 *tj, float *tx) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n_keys; k += (i64)gridDim.x * blockDim.x) {
        i32 a, b; u64 ord; float v;
        if (!vw.get(k, a, b, ord, v)) continue;
        a = frag_index[a]; b = frag_index[b];
        if (a < 0 || b < 0) continue;
        i32 p = indptr[a] + atomicAdd(&cursor[a], 1);
        tj[p] = b; tx[p] = v;
        p = indptr[b] + atomicAdd(&cursor[b], 1);
        tj[p] = a; tx[p] = v;
    }
}
__global__ __launch_bounds__(256) void k_fill_diag(i32 shape, const i32 *__restrict__ indptr, i32 *cursor, i32 *tj, float *tx) {
    for (i32 r = blockIdx.x * blockDim.x + threadIdx.x; r < shape; r += gridDim.x * blockDim.x) {
        const i32 p = indptr[r] + atomicAdd(&cursor[r], 1);
        tj[p] = r; tx[p] = 1.0f;                                    // self loops :362-364
    }
}
// ... then each row is put in column order with an LDS bitmap rank (columns of a row are unique):
// the same no-sort trick as the SpGEMM output.
__global__ __launch_bounds__(256) void k_sort_rows(i32 shape, i32 W, const i32 *__restrict__ indptr, const i32 *__restrict__ tj,
                                                   const float *__restrict__ tx, i32 *__restrict__ oj, float *__restrict__ ox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem, *prefix = bitmap + W, *scratch = prefix + W;
    const int tid = threadIdx.x;
    for (i32 row = blockIdx.x; row < shape; row += gridDim.x) {
        const i32 b = indptr[row], e = indptr[row + 1];
        if (e - b <= 1) {
            if (tid == 0 && e > b) { oj[b] = tj[b]; ox[b] = tx[b]; }
            continue;
        }
        for (i32 w = tid; w < W; w += 256) bitmap[w] = 0;
        __syncthreads();
        for (i32 p = b + tid; p < e; p += 256) atomicOr(&bitmap[tj[p] >> 5], 1u << (tj[p] & 31));
        __syncthreads();
        // exclusive popcount prefix over the words (serial chunk per thread + 256-entry scan)
        const i32 per = (W + 255) / 256, w0 = tid * per, w1 = min(W, w0 + per);
        u32 local = 0;
        for (i32 w = w0; w < w1; ++w) local += __popc(bitmap[w]);
        scratch[tid] = local;
        __syncthreads();
        if (tid < 64) {
            u32 v0 = scratch[tid * 4], v1 = scratch[tid * 4 + 1], v2 = scratch[tid * 4 + 2], v3 = scratch[tid * 4 + 3];
            u32 s = v0 + v1 + v2 + v3, incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                u32 t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            u32 ex = incl - s;
            scratch[tid * 4] = ex; scratch[tid * 4 + 1] = ex + v0; scratch[tid * 4 + 2] = ex + v0 + v1; scratch[tid * 4 + 3] = ex + v0 + v1 + v2;
        }
        __syncthreads();
        u32 run = scratch[tid];
        for (i32 w = w0; w < w1; ++w) { prefix[w] = run; run += __popc(bitmap[w]); }
        __syncthreads();
        for (i32 p = b + tid; p < e; p += 256) {
            const i32 c = tj[p];
            const i32 r = (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
            oj[b + r] = c;
            ox[b + r] = tx[p];
        }
        __syncthreads();
    }
}

inline unsigned grid_for(u64 n) {
    u64 b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

template <class View>
int build_matrix(const View &vw, i64 n_keys, i32 n_frag, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                 i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out) {
    DevBuf<unsigned char> in_set;
    if (in_set.alloc((size_t)n_frag)) return 1;
    HHX_HIP(hipMemcpyAsync(in_set.p, in_set_host, (size_t)n_frag, hipMemcpyHostToDevice, g_stream));
    DevBuf<unsigned long long> first_pos;
    DevBuf<i32> frag_index;
    DevBuf<unsigned int> nl;
    if (first_pos.alloc((size_t)n_frag) || frag_index.alloc((size_t)n_frag) || nl.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(first_pos.p, 0xff, sizeof(unsigned long long) * (size_t)n_frag, g_stream));
    HHX_HIP(hipMemsetAsync(nl.p, 0, sizeof(unsigned int), g_stream));
    if (n_keys) {
        k_first_pos<View><<<grid_for((u64)n_keys), 256, 0, g_stream>>>(vw, n_keys, in_set.p, first_pos.p);
        HHX_LAUNCH_CHECK();
    }
    k_rank_first<<<(unsigned)((n_frag + 255) / 256), 256, 0, g_stream>>>(n_frag, first_pos.p, frag_index.p, nl.p);
    HHX_LAUNCH_CHECK();
    unsigned int n_linked = 0;
    HHX_HIP(hipMemcpyAsync(&n_linked, nl.p, sizeof n_linked, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (n_rest < 0) {                                           // every link-less member of frag_set follows the linked ones
        i64 members = 0;
        for (i32 f = 0; f < n_frag; ++f) members += in_set_host[f] != 0;
        n_rest = (i32)(members - (i64)n_linked);
    }
    const i64 shape64 = (i64)n_linked + n_rest;
    if (shape64 > INT32_MAX) return fail("matrix order exceeds int32");
    const i32 shape = (i32)shape64;
    DevBuf<i32> cnt, indptr, cursor;
    if (cnt.alloc((size_t)shape + 1) || indptr.alloc((size_t)shape + 2) || cursor.alloc((size_t)shape + 1)) return 1;
    k_init_counts<<<grid_for((u64)shape + 1), 256, 0, g_stream>>>(shape, cnt.p, add_self_loops ? 1 : 0);
    HHX_LAUNCH_CHECK();
    if (n_keys) {
        k_row_counts<View><<<grid_for((u64)n_keys), 256, 0, g_stream>>>(vw, n_keys, frag_index.p, cnt.p);
        HHX_LAUNCH_CHECK();
    }
    i64 nnz = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, indptr.p, shape, &nnz));
    hhx_csr *m = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(shape, shape, nnz, &m));
    DevBuf<i32> tj;
    DevBuf<float> tx;
    if (tj.alloc((size_t)nnz) || tx.alloc((size_t)nnz)) { hhx_csr_free(m); return 1; }
    hipError_t e = hipMemsetAsync(cursor.p, 0, sizeof(i32) * ((size_t)shape + 1), g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(m->indptr.p, indptr.p, sizeof(i32) * ((size_t)shape + 1), hipMemcpyDeviceToDevice, g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("dict_to_matrix: %s", hipGetErrorString(e)); }
    if (n_keys) k_fill<View><<<grid_for((u64)n_keys), 256, 0, g_stream>>>(vw, n_keys, frag_index.p, indptr.p, cursor.p, tj.p, tx.p);
    if (add_self_loops && shape) k_fill_diag<<<grid_for((u64)shape), 256, 0, g_stream>>>(shape, indptr.p, cursor.p, tj.p, tx.p);
    const i32 W = (shape + 31) / 32;
    const size_t lds = (size_t)W * 8 + 256 * 4;
    if (lds > 160 * 1024) { hhx_csr_free(m); return fail("dict_to_matrix: matrix order %d exceeds the LDS bitmap capacity", shape); }
    static int attr_set = -1;           // the attribute is per device: keyed on the current ordinal
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        (void)hipFuncSetAttribute((const void *)k_sort_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = attr_dev;
    }
    if (shape) k_sort_rows<<<(unsigned)std::min<i64>(shape, 256 * 8), 256, lds, g_stream>>>(shape, W, indptr.p, tj.p, tx.p, m->indices.p, m->data.p);
    e = hipGetLastError();
    if (e == hipSuccess && frag_index_host)
        e = hipMemcpyAsync(frag_index_host, frag_index.p, sizeof(i32) * (size_t)n_frag, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("dict_to_matrix: %s", hipGetErrorString(e)); }
    if (n_linked_out) *n_linked_out = (i32)n_linked;
    *out = m;
    return 0;
}

template <class T>
int upload(DevBuf<T> &d, const T *h, size_t n) {
    if (d.alloc(n)) return 1;
    if (n) HHX_HIP(hipMemcpyAsync(d.p, h, n * sizeof(T), hipMemcpyHostToDevice, g_stream));
    return 0;
}

}  // namespace

extern "C" int hhx_dict_to_matrix(i64 n_keys, const i32 *frag_i, const i32 *frag_j, const double *value, int on_device,
                                  i32 n_frag, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                                  i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out) {
    if (!out || !in_set_host || n_frag <= 0 || n_keys < 0) return fail("hhx_dict_to_matrix: bad argument");
    DevBuf<i32> sfi, sfj;
    DevBuf<double> sval;
    if (!on_device && n_keys) {
        if (upload(sfi, frag_i, (size_t)n_keys) || upload(sfj, frag_j, (size_t)n_keys) || upload(sval, value, (size_t)n_keys)) return 1;
        frag_i = sfi.p; frag_j = sfj.p; value = sval.p;
    }
    const RowsView vw{frag_i, frag_j, value};
    return build_matrix(vw, n_keys, n_frag, in_set_host, n_rest, add_self_loops, frag_index_host, n_linked_out, out);
}

// ================================================================================================
// Fast path of hhx_ingest_link_matrix: the matrix is built by PARTITIONING directed entries by row.
// Every flank key (i, j, count, first ordinal) with both ends in frag_set yields two 16-byte entries
//     w0 = row << 29 | column-fragment,   w1 = (2 * ordinal + side) << 31 | count
// which the radix partition of hhx_partition.h groups by row (bucket = fragment id).  After that everything
// is row-local and streaming: first position and length of a row are a wave reduction over its entries (no
// per-fragment atomics), and a workgroup per row maps the column fragments to matrix indices, sorts them
// with the LDS bitmap rank and writes the CSR row.  The generic path above needs 6 random-access passes with
// L2 atomics over the keys (55 ms at 165 M keys; its scattered fill wrote 12x the bytes it stored).
namespace {

struct SrcDirected {
    typedef u64 w1_t;
    static constexpr bool MARK = false;
    const u64 *key, *ord_flank;
    const u32 *fl;
    const unsigned char *in_set;
    __device__ __forceinline__ bool get(i64 idx, u64 &w0, u64 &w1) const {
        const i64 k = idx >> 1;
        const u64 ord = ord_flank[k];
        if (ord == NO_ORD) return false;
        const u64 ky = key[k];
        const u32 i = (u32)(ky >> ID_BITS), j = (u32)(ky & ID_MASK);
        if (!in_set[i] || !in_set[j]) return false;
        const u32 side = (u32)(idx & 1);
        const u32 a = side ? j : i, b = side ? i : j;
        w0 = ((u64)a << ID_BITS) | (u64)b;
        w1 = ((2 * ord + side) << 31) | (u64)fl[k];
        return true;
    }
};
struct DigRow {
    __device__ __forceinline__ u32 operator()(u64 w0) const { return (u32)(w0 >> ID_BITS); }
};

// PACKED entries: fragment ids below 2^20 and counts below 2^24 (every real link table) fit ONE 64-bit word
//     w0 = row << 44 | column-fragment << 24 | count
// so the two partition levels and the row emit move 8 bytes per entry instead of 16 and the scatter stages 14 entries
// per thread.  The first positions the 16-byte entries carry in w1 are taken by the level-1 count pass as a side
// effect of reading the table (SrcDirectedPacked::first_pos).
constexpr int PK_ID_BITS = 20, PK_CNT_BITS = 24;
constexpr u32 PK_ID_MASK = (1u << PK_ID_BITS) - 1, PK_CNT_MASK = (1u << PK_CNT_BITS) - 1;
__device__ __forceinline__ u32 xcc_id() {            // the XCD this wave runs on: s_getreg_b32 HW_REG_XCC_ID (register 20, bits 3:0)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
#else
    return 0;
#endif
}
struct SrcDirectedPacked {
    typedef NoPayload w1_t;
    static constexpr bool MARK = true;      // k_part_count calls mark_load / mark_apply once per record (level 1 only)
    const u64 *key, *ord_flank;
    const u32 *fl;
    const unsigned char *in_set;
    unsigned long long *first_pos;          // [N_XCC][n_frag]; set in the object the level-1 count pass reads through, null otherwise
    unsigned int *too_big;
    i32 n_frag;
    // extra: the entry's position in the insertion order, 2 * ordinal + side (all ones: its count does not fit PK_CNT_BITS)
    __device__ __forceinline__ bool get_marked(i64 idx, u64 &w0, u64 &extra) const {
        const i64 k = idx >> 1;
        const u64 ord = ord_flank[k];
        if (ord == NO_ORD) return false;
        const u64 ky = key[k];
        const u32 i = (u32)(ky >> ID_BITS), j = (u32)(ky & ID_MASK);
        if (!in_set[i] || !in_set[j]) return false;
        const u32 side = (u32)(idx & 1);
        const u32 a = side ? j : i, b = side ? i : j;
        const u32 c = fl[k];
        w0 = ((u64)a << (PK_ID_BITS + PK_CNT_BITS)) | ((u64)b << PK_CNT_BITS) | (u64)(c & PK_CNT_MASK);
        extra = c > PK_CNT_MASK ? ~0ull : 2 * ord + side;
        return true;
    }
    __device__ __forceinline__ bool get(i64 idx, u64 &w0, NoPayload &) const {
        u64 extra;
        return get_marked(idx, w0, extra);
    }
    // First position of the row fragment: an atomic min that almost never fires, because the current minimum is read first
    // and the table is walked in hash order (a minimum settles after ~ln(entries) updates).  For that read to see the
    // updates it must be served by the cache the atomics pass through: the per-XCD L2s are not coherent with each other, so
    // every XCD keeps ITS OWN copy of the table (first_pos[xcc][n_frag], merged by k_min_over_xcc afterwards) — an atomic
    // drops the line from the issuing XCD's L2, the next L2-served (sc1: past the CU's L1) load fetches the new value.
    // With one shared table each XCD's L2 kept the initial ~0 and every lane fired: 3 ms of fabric atomics per 330 M entries.
    __device__ __forceinline__ unsigned long long *my_table() const {
        return first_pos + (size_t)xcc_id() * (size_t)n_frag;
    }
    __device__ __forceinline__ u64 mark_load(u64 w0) const {
        if (!first_pos) return 0;
        return __hip_atomic_load(&my_table()[(u32)(w0 >> (PK_ID_BITS + PK_CNT_BITS)) & PK_ID_MASK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void mark_apply(u64 w0, u64 pos, u64 seen) const {
        if (!first_pos) return;
        if (pos == ~0ull) *too_big = 1u;                                            // 16-byte entries then (the caller starts over)
        else if (pos < seen) atomicMin(&my_table()[(u32)(w0 >> (PK_ID_BITS + PK_CNT_BITS)) & PK_ID_MASK], (unsigned long long)pos);
    }
};
constexpr int N_XCC = 8;
__global__ __launch_bounds__(256) void k_min_over_xcc(i32 n_frag, const unsigned long long *__restrict__ per_xcc, unsigned long long *__restrict__ out) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x) {
        unsigned long long m = ~0ull;
#pragma unroll
        for (int x = 0; x < N_XCC; ++x) m = min(m, per_xcc[(size_t)x * n_frag + f]);
        out[f] = m;
    }
}
struct DigRowPacked {
    __device__ __forceinline__ u32 operator()(u64 w0) const { return (u32)(w0 >> (PK_ID_BITS + PK_CNT_BITS)) & PK_ID_MASK; }
};
__global__ __launch_bounds__(256) void k_len_from_base(i32 n_frag, const i64 *__restrict__ base, const i32 *__restrict__ frag_index, i32 *__restrict__ cnt) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        if (frag_index[f] >= 0) cnt[frag_index[f]] += (i32)(base[f + 1] - base[f]);          // one writer per index
}
// matrix index = rank of the first position: (position, fragment) pairs through the stable radix sort; positions are
// distinct, fragments without entries (~0) sort behind every real one.  (k_rank_first's all-pairs count is n^2:
// 1.5 ms at 100k fragments, 6 ms at 200k.)
__global__ __launch_bounds__(256) void k_iota_u64(i64 n, u64 *out) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) out[k] = (u64)k;
}
__global__ __launch_bounds__(256) void k_index_from_sorted(i64 n, const u64 *__restrict__ pos_sorted, const u64 *__restrict__ frag_sorted,
                                                           i32 *__restrict__ frag_index, unsigned int *n_linked) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const bool linked = pos_sorted[k] != ~0ull;
        frag_index[frag_sorted[k]] = linked ? (i32)k : -1;
        if (linked && (k + 1 == n || pos_sorted[k + 1] == ~0ull)) *n_linked = (unsigned int)(k + 1);
    }
}
int rank_first_positions(i32 n_frag, const unsigned long long *first_pos, u64 pos_limit, i32 *frag_index, unsigned int *n_linked_dev) {
    if (n_frag <= 4096) {
        k_rank_first<<<(unsigned)((n_frag + 255) / 256), 256, 0, g_stream>>>(n_frag, first_pos, frag_index, n_linked_dev);
        HHX_LAUNCH_CHECK();
        return 0;
    }
    int bits = 1;
    while (bits < 64 && (pos_limit + 1) >> bits) ++bits;              // every real position < 2^bits - 1: ~0 keeps the top slot
    DevBuf<u64> iota, ks, vs;
    if (iota.alloc((size_t)n_frag) || ks.alloc((size_t)n_frag) || vs.alloc((size_t)n_frag)) return 1;
    k_iota_u64<<<(unsigned)((n_frag + 255) / 256), 256, 0, g_stream>>>(n_frag, iota.p);
    HHX_LAUNCH_CHECK();
    HHX_TRY(stable_sort_pairs_u64((const u64 *)first_pos, ks.p, iota.p, vs.p, n_frag, bits));
    k_index_from_sorted<<<(unsigned)((n_frag + 255) / 256), 256, 0, g_stream>>>(n_frag, ks.p, vs.p, frag_index, n_linked_dev);
    HHX_LAUNCH_CHECK();
    HHX_HIP(hipStreamSynchronize(g_stream));                           // ks / vs die here
    return 0;
}

// first position (min over the row's entries) and length of every row; one wave per row
__global__ __launch_bounds__(256) void k_row_first(i32 n_frag, const i64 *__restrict__ base, const u64 *__restrict__ w1,
                                                   unsigned long long *__restrict__ first_pos, i32 *__restrict__ row_len) {
    const int lane = lane_id();
    for (i32 a = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; a < n_frag; a += gridDim.x * 4) {
        const i64 b = base[a], e = base[a + 1];
        unsigned long long m = ~0ull;
        for (i64 p = b + lane; p < e; p += HHX_WAVE) m = min(m, (unsigned long long)(w1[p] >> 31));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned long long)__shfl_down((long long)m, o, HHX_WAVE));
        if (lane == 0) { first_pos[a] = m; row_len[a] = (i32)(e - b); }
    }
}
__global__ __launch_bounds__(256) void k_len_by_index(i32 n_frag, const i32 *__restrict__ frag_index, const i32 *__restrict__ row_len,
                                                      i32 *__restrict__ cnt) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        if (frag_index[f] >= 0) cnt[frag_index[f]] += row_len[f];           // one writer per index
}
template <bool PACKED>
__device__ __forceinline__ u32 entry_col(u64 w0) { return PACKED ? ((u32)(w0 >> PK_CNT_BITS) & PK_ID_MASK) : (u32)(w0 & ID_MASK); }
template <bool PACKED>
__device__ __forceinline__ float entry_val(u64 w0, const u64 *__restrict__ w1, i64 p) {
    return PACKED ? (float)((u32)w0 & PK_CNT_MASK) : (float)(u32)(w1[p] & 0x7fffffffu);
}
// One workgroup per linked fragment: CSR row in column order (LDS bitmap rank; columns of a row are unique).
// The rows are short (3.3k entries on average at C3), so a row is a chain of dependent round trips — entries, index
// gather, barriers, stores — and the kernel is latency bound: the entries of the NEXT row of the workgroup are loaded
// into registers before the current row's gather / rank / store phases begin.
constexpr int EMIT_T = 512, EMIT_R = 8;
template <bool PACKED>
__global__ __launch_bounds__(EMIT_T) void k_row_emit(i32 n_frag, i32 W, int self_loop, const i64 *__restrict__ base, const u64 *__restrict__ w0,
                                                     const u64 *__restrict__ w1, const i32 *__restrict__ frag_index,
                                                     const i32 *__restrict__ indptr, i32 *__restrict__ oj, float *__restrict__ ox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem, *prefix = bitmap + W, *wsum = prefix + W;       // wsum[EMIT_T / 64]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / HHX_WAVE;
    i32 a = blockIdx.x;
    if (a >= n_frag) return;
    // registers of the row about to be processed
    i32 r = frag_index[a];
    i64 b = base[a], e = r < 0 ? b : base[a + 1];
    u64 t0[EMIT_R];
#pragma unroll
    for (int u = 0; u < EMIT_R; ++u) { const i64 p = b + tid + (i64)u * EMIT_T; t0[u] = p < e ? w0[p] : 0; }
    // the loop is entered with t0 DEFINED (no load pending on it), and inside it the prefetched tn is pinned before the row's
    // stores are issued: the waitcnt pass is static and vmcnt completes in order, so a pending load at the loop entry puts a
    // vmcnt(0) in front of every use of t0, and a wait for tn placed after the stores drains the stores, every row
#pragma unroll
    for (int u = 0; u < EMIT_R; ++u) asm volatile("" : "+v"(t0[u]));
    for (;;) {
        const i32 an = a + gridDim.x;
        i32 rn = -1;
        i64 bn = 0, en = 0;
        u64 tn[EMIT_R];
        if (an < n_frag) { rn = frag_index[an]; bn = base[an]; en = rn < 0 ? bn : base[an + 1]; }
#pragma unroll
        for (int u = 0; u < EMIT_R; ++u) { const i64 p = bn + tid + (i64)u * EMIT_T; tn[u] = p < en ? w0[p] : 0; }
        if (r >= 0) {
            const i32 ob = indptr[r];
            for (i32 w = tid; w < W; w += EMIT_T) bitmap[w] = 0;
            i32 cc[EMIT_R];
#pragma unroll
            for (int u = 0; u < EMIT_R; ++u) { const i64 p = b + tid + (i64)u * EMIT_T; cc[u] = p < e ? frag_index[entry_col<PACKED>(t0[u])] : -1; }
            lds_barrier();
#pragma unroll
            for (int u = 0; u < EMIT_R; ++u) if (cc[u] >= 0) atomicOr(&bitmap[cc[u] >> 5], 1u << (cc[u] & 31));
            for (i64 p = b + tid + (i64)EMIT_R * EMIT_T; p < e; p += EMIT_T) { const i32 c = frag_index[entry_col<PACKED>(w0[p])]; atomicOr(&bitmap[c >> 5], 1u << (c & 31)); }
            if (self_loop && tid == 0) atomicOr(&bitmap[r >> 5], 1u << (r & 31));
            lds_barrier();
            // exclusive prefix of the word popcounts: a contiguous chunk of words per thread, wave scan, wave totals in LDS
            const i32 per = (W + EMIT_T - 1) / EMIT_T, wa = min(W, tid * per), wb = min(W, wa + per);
            u32 local = 0;
            for (i32 w = wa; w < wb; ++w) local += __popc(bitmap[w]);
            u32 incl = local;
#pragma unroll
            for (int o = 1; o < HHX_WAVE; o <<= 1) {
                const u32 t = __shfl_up(incl, o, HHX_WAVE);
                if (lane >= o) incl += t;
            }
            if (lane == HHX_WAVE - 1) wsum[wave] = incl;
            lds_barrier();
            u32 run = incl - local;
            for (int w = 0; w < wave; ++w) run += wsum[w];
            for (i32 w = wa; w < wb; ++w) { prefix[w] = run; run += __popc(bitmap[w]); }
            lds_barrier();
#pragma unroll
            for (int u = 0; u < EMIT_R; ++u) asm volatile("" : "+v"(tn[u]));
            asm volatile("" : "+v"(rn)); asm volatile("" : "+v"(bn)); asm volatile("" : "+v"(en));
#pragma unroll
            for (int u = 0; u < EMIT_R; ++u)
                if (cc[u] >= 0) {
                    const i32 c = cc[u];
                    const i32 k = (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
                    oj[ob + k] = c;
                    ox[ob + k] = entry_val<PACKED>(t0[u], w1, b + tid + (i64)u * EMIT_T);
                }
            for (i64 p = b + tid + (i64)EMIT_R * EMIT_T; p < e; p += EMIT_T) {
                const u64 t = w0[p];
                const i32 c = frag_index[entry_col<PACKED>(t)];
                const i32 k = (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
                oj[ob + k] = c;
                ox[ob + k] = entry_val<PACKED>(t, w1, p);
            }
            if (self_loop && tid == 0) {
                const i32 k = (i32)(prefix[r >> 5] + __popc(bitmap[r >> 5] & ((1u << (r & 31)) - 1u)));
                oj[ob + k] = r;
                ox[ob + k] = 1.0f;                                  // self loops :362-364
            }
            lds_barrier();
        } else {
#pragma unroll
            for (int u = 0; u < EMIT_R; ++u) asm volatile("" : "+v"(tn[u]));
            asm volatile("" : "+v"(rn)); asm volatile("" : "+v"(bn)); asm volatile("" : "+v"(en));
        }
        if (an >= n_frag) break;
        a = an; r = rn; b = bn; e = en;
#pragma unroll
        for (int u = 0; u < EMIT_R; ++u) t0[u] = tn[u];
    }
}
// link-less members of frag_set: a unit self loop only
__global__ __launch_bounds__(256) void k_rest_rows(i32 r0, i32 shape, const i32 *__restrict__ indptr, i32 *__restrict__ oj, float *__restrict__ ox) {
    for (i32 r = r0 + blockIdx.x * blockDim.x + threadIdx.x; r < shape; r += gridDim.x * blockDim.x) { oj[indptr[r]] = r; ox[indptr[r]] = 1.0f; }
}

// PACKED: 8-byte entries (see SrcDirectedPacked); returns -2 when a count does not fit 24 bits (the caller takes the 16-byte path)
template <bool PACKED>
int link_matrix_partitioned(const LinkRun *run, i32 n_frag, u64 ord_limit, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                            i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out) {
    typedef typename std::conditional<PACKED, NoPayload, u64>::type W1;
    DevBuf<unsigned char> in_set;
    if (in_set.alloc((size_t)n_frag)) return 1;
    HHX_HIP(hipMemcpyAsync(in_set.p, in_set_host, (size_t)n_frag, hipMemcpyHostToDevice, g_stream));
    int row_bits = 0;
    while (((i64)1 << row_bits) < n_frag) ++row_bits;
    DevBuf<unsigned long long> first_pos;
    DevBuf<i32> frag_index, row_len;
    DevBuf<unsigned int> nl;
    if (first_pos.alloc((size_t)n_frag) || frag_index.alloc((size_t)n_frag) || nl.alloc(2)) return 1;
    HHX_HIP(hipMemsetAsync(nl.p, 0, 2 * sizeof(unsigned int), g_stream));
    Partitioned<W1> part;
    static const int level_bits = getenv("HHX_D2M_LBITS") ? atoi(getenv("HHX_D2M_LBITS")) : 9;
    if constexpr (PACKED) {
        DevBuf<unsigned long long> per_xcc;
        if (per_xcc.alloc((size_t)N_XCC * (size_t)n_frag)) return 1;
        HHX_HIP(hipMemsetAsync(per_xcc.p, 0xff, sizeof(unsigned long long) * (size_t)N_XCC * (size_t)n_frag, g_stream));
        const SrcDirectedPacked src{run->key.p, run->ord_flank.p, run->fl.p, in_set.p, nullptr, nullptr, n_frag};
        const SrcDirectedPacked marking{run->key.p, run->ord_flank.p, run->fl.p, in_set.p, per_xcc.p, nl.p + 1, n_frag};
        HHX_TRY(partition_records(src, DigRowPacked(), 2 * run->n, row_bits, level_bits, &part, "d2m", &marking));
        k_min_over_xcc<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, per_xcc.p, first_pos.p);
        HHX_LAUNCH_CHECK();
        unsigned int too_big = 0;
        HHX_HIP(hipMemcpyAsync(&too_big, nl.p + 1, sizeof too_big, hipMemcpyDeviceToHost, g_stream));
        HHX_HIP(hipStreamSynchronize(g_stream));
        if (too_big) return -2;
    } else {
        const SrcDirected src{run->key.p, run->ord_flank.p, run->fl.p, in_set.p};
        HHX_TRY(partition_records(src, DigRow(), 2 * run->n, row_bits, level_bits, &part, "d2m"));
        if (row_len.alloc((size_t)n_frag)) return 1;
        if (part.n_valid == 0) {
            HHX_HIP(hipMemsetAsync(first_pos.p, 0xff, sizeof(unsigned long long) * (size_t)n_frag, g_stream));
            HHX_HIP(hipMemsetAsync(row_len.p, 0, sizeof(i32) * (size_t)n_frag, g_stream));
        } else {
            KTimer kt("d2m_row_first");
            k_row_first<<<grid_for((u64)n_frag * 64), 256, 0, g_stream>>>(n_frag, part.base.p, part.w1.p, first_pos.p, row_len.p);
        }
        HHX_LAUNCH_CHECK();
    }
    { KTimer kt("d2m_rank");
    HHX_TRY(rank_first_positions(n_frag, first_pos.p, 2 * ord_limit + 1, frag_index.p, nl.p)); }
    unsigned int n_linked = 0;
    HHX_HIP(hipMemcpyAsync(&n_linked, nl.p, sizeof n_linked, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (n_rest < 0) {
        i64 members = 0;
        for (i32 f = 0; f < n_frag; ++f) members += in_set_host[f] != 0;
        n_rest = (i32)(members - (i64)n_linked);
    }
    const i64 shape64 = (i64)n_linked + n_rest;
    if (shape64 > INT32_MAX) return fail("matrix order exceeds int32");
    const i32 shape = (i32)shape64;
    DevBuf<i32> cnt, indptr;
    if (cnt.alloc((size_t)shape + 1) || indptr.alloc((size_t)shape + 2)) return 1;
    k_init_counts<<<grid_for((u64)shape + 1), 256, 0, g_stream>>>(shape, cnt.p, add_self_loops ? 1 : 0);
    if (part.n_valid) {
        if (PACKED) k_len_from_base<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, part.base.p, frag_index.p, cnt.p);
        else k_len_by_index<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, frag_index.p, row_len.p, cnt.p);
    }
    HHX_LAUNCH_CHECK();
    i64 nnz = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, indptr.p, shape, &nnz));
    hhx_csr *m = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(shape, shape, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->indptr.p, indptr.p, sizeof(i32) * ((size_t)shape + 1), hipMemcpyDeviceToDevice, g_stream);
    const i32 W = (shape + 31) / 32;
    const size_t lds = (size_t)W * 8 + (EMIT_T / HHX_WAVE) * 4;
    if (lds > 160 * 1024) { hhx_csr_free(m); return fail("link matrix: order %d exceeds the LDS bitmap capacity", shape); }
    static int attr_set = -1;           // the attribute is per device: keyed on the current ordinal
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        (void)hipFuncSetAttribute((const void *)k_row_emit<PACKED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = attr_dev;
    }
    if (e == hipSuccess && part.n_valid) {
        KTimer kt("d2m_emit");
        k_row_emit<PACKED><<<(unsigned)std::min<i64>(n_frag, 256 * 6), EMIT_T, lds, g_stream>>>(n_frag, W, add_self_loops, part.base.p, part.w0.p,
                                                                                          (const u64 *)part.w1.p, frag_index.p, indptr.p, m->indices.p, m->data.p);
    }
    if (e == hipSuccess && add_self_loops && shape > (i32)n_linked)
        k_rest_rows<<<grid_for((u64)(shape - (i32)n_linked)), 256, 0, g_stream>>>((i32)n_linked, shape, indptr.p, m->indices.p, m->data.p);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && frag_index_host)
        e = hipMemcpyAsync(frag_index_host, frag_index.p, sizeof(i32) * (size_t)n_frag, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("link matrix: %s", hipGetErrorString(e)); }
    if (n_linked_out) *n_linked_out = (i32)n_linked;
    *out = m;
    return 0;
}

}  // namespace

int hhx_link_matrix_from_run(const LinkRun *run, i32 n_frag, u64 ord_limit, const uint8_t *in_set_host, i32 n_rest, int add_self_loops,
                             i32 *frag_index_host, i32 *n_linked_out, hhx_csr **out) {
    if (!out || !in_set_host || n_frag <= 0) return fail("hhx_ingest_link_matrix: bad argument");
    // packed entries hold 2*ordinal+side in 33 bits and the count in 31: true whenever < 2^31 pairs were seen
    if (run && run->n && n_frag <= (1 << 20) && ord_limit <= ((u64)1 << 31) && !getenv("HHX_D2M_GENERIC")) {
        if (!getenv("HHX_D2M_WIDE")) {                           // 8-byte entries unless a count needs more than 24 bits
            const int rc = link_matrix_partitioned<true>(run, n_frag, ord_limit, in_set_host, n_rest, add_self_loops, frag_index_host, n_linked_out, out);
            if (rc != -2) return rc;
        }
        return link_matrix_partitioned<false>(run, n_frag, ord_limit, in_set_host, n_rest, add_self_loops, frag_index_host, n_linked_out, out);
    }
    const RunView vw{run ? run->key.p : nullptr, run ? run->ord_flank.p : nullptr, run ? run->fl.p : nullptr};
    return build_matrix(vw, run ? run->n : 0, n_frag, in_set_host, n_rest, add_self_loops, frag_index_host, n_linked_out, out);
}

// ================================================================================================
// Multi-GPU build of the link matrix (haphic_amd/sharded.py: build_link_matrix_sharded).  Every rank holds the
// aggregated table of ITS chunk of the pair stream (ordinals are global stream ordinals).  dict_to_matrix's index of
// a fragment is the rank of its first position 2 * ordinal + side over the whole stream, and a first position is a
// minimum — so the ranks all-reduce(min) one int64 per fragment, rank it identically, and then exchange matrix
// ENTRIES (row index, column index, count) by row owner instead of gathering every rank's whole table:
//   hhx_shard_create     this rank's directed entries partitioned by row fragment (same pipeline as the 1-GPU build)
//   hhx_shard_first      first position of every fragment in this chunk (INT64_MAX: none)      -> all-reduce(min)
//   hhx_rank_first       matrix index of every fragment from the reduced positions (replicated)
//   hhx_shard_emit       the entries rewritten as (row << 29 | column, count), sorted by matrix row -> all-to-all(v)
//   hhx_rows_from_entries  the owner groups what it received by row, adds up the counts of equal (row, column)
//                        coming from different chunks, and writes its CSR row block (self loops, link-less rows)
struct hhx_shard {
    i32 n_frag = 0;
    hhx::Partitioned<u64> part;
    hhx::DevBuf<long long> first;          // [n_frag]
    hhx::DevBuf<i32> row_len;              // [n_frag] entries of the fragment in this chunk
    hhx::DevBuf<u64> out_w0, out_w1;
    hhx::DevBuf<unsigned char> in_set;
};

namespace {

constexpr long long SHARD_NONE = INT64_MAX;

__global__ __launch_bounds__(256) void k_first_signed(i32 n_frag, const unsigned long long *__restrict__ first_pos, long long *__restrict__ out) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        out[f] = first_pos[f] == ~0ull ? SHARD_NONE : (long long)first_pos[f];
}
__global__ __launch_bounds__(256) void k_len_by_row(i32 n_frag, const i32 *__restrict__ frag_index, const i32 *__restrict__ row_len,
                                                    i64 *__restrict__ len_by_row) {
    for (i32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frag; f += gridDim.x * blockDim.x)
        if (frag_index[f] >= 0) len_by_row[frag_index[f]] = row_len[f];             // one writer per row
}
// one wave per fragment: its entries move to the slot of its matrix row, fragment ids become matrix indices
__global__ __launch_bounds__(256) void k_emit_by_row(i32 n_frag, const i64 *__restrict__ base, const u64 *__restrict__ w0, const u64 *__restrict__ w1,
                                                     const i32 *__restrict__ frag_index, const i64 *__restrict__ row_off,
                                                     u64 *__restrict__ o0, u64 *__restrict__ o1) {
    const int lane = lane_id();
    for (i32 a = blockIdx.x * 4 + threadIdx.x / HHX_WAVE; a < n_frag; a += gridDim.x * 4) {
        const i32 r = frag_index[a];
        const i64 b = base[a], e = base[a + 1];
        if (r < 0 || b == e) continue;
        const i64 d = row_off[r];
        for (i64 p = b + lane; p < e; p += HHX_WAVE) {
            o0[d + (p - b)] = ((u64)(u32)r << ID_BITS) | (u64)(u32)frag_index[(u32)(w0[p] & ID_MASK)];
            o1[d + (p - b)] = w1[p] & 0x7fffffffull;
        }
    }
}
struct DigLocalRow {
    u32 r0;
    __device__ __forceinline__ u32 operator()(u64 w0) const { return (u32)(w0 >> ID_BITS) - r0; }
};
// LDS bitmap of the distinct columns of one row; returns nothing, leaves bitmap[] filled (all threads must call)
__device__ __forceinline__ void mark_columns(u32 *bitmap, i32 W, const u64 *__restrict__ w0, i64 b, i64 e, i32 self_col) {
    for (i32 w = threadIdx.x; w < W; w += 256) bitmap[w] = 0;
    __syncthreads();
    for (i64 p = b + threadIdx.x; p < e; p += 256) { const u32 c = (u32)(w0[p] & ID_MASK); atomicOr(&bitmap[c >> 5], 1u << (c & 31)); }
    if (self_col >= 0 && threadIdx.x == 0) atomicOr(&bitmap[self_col >> 5], 1u << (self_col & 31));
    __syncthreads();
}
// distinct columns per local row (+ the self loop)
__global__ __launch_bounds__(256) void k_rows_distinct(i32 n_local, i32 r0, i32 W, int self_loop, const i64 *__restrict__ base, const u64 *__restrict__ w0,
                                                       i32 *__restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem;
    i32 *wsum = (i32 *)(bitmap + W);                             // no static LDS: the 160 KB attribute needs it all dynamic
    for (i32 a = blockIdx.x; a < n_local; a += gridDim.x) {
        const i64 b = base ? base[a] : 0, e = base ? base[a + 1] : 0;
        if (b == e) { if (threadIdx.x == 0) cnt[a] = self_loop ? 1 : 0; continue; }
        mark_columns(bitmap, W, w0, b, e, self_loop ? r0 + a : -1);
        i32 c = 0;
        for (i32 w = threadIdx.x; w < W; w += 256) c += __popc(bitmap[w]);
        c = wave_sum_i32(c);
        if (lane_id() == 0) wsum[threadIdx.x / HHX_WAVE] = c;
        __syncthreads();
        if (threadIdx.x == 0) cnt[a] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}
// CSR row block: columns in order (bitmap rank), counts of equal columns added up as integers, then cast to float
__global__ __launch_bounds__(256) void k_rows_merge_emit(i32 n_local, i32 r0, i32 W, int self_loop, const i64 *__restrict__ base, const u64 *__restrict__ w0,
                                                         const u64 *__restrict__ w1, const i32 *__restrict__ indptr, i32 *__restrict__ oj,
                                                         float *__restrict__ ox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *bitmap = (u32 *)smem, *prefix = bitmap + W, *scratch = prefix + W;
    const int tid = threadIdx.x;
    u32 *oxu = reinterpret_cast<u32 *>(ox);
    for (i32 a = blockIdx.x; a < n_local; a += gridDim.x) {
        const i64 b = base ? base[a] : 0, e = base ? base[a + 1] : 0;
        const i32 ob = indptr[a], oe = indptr[a + 1], r = r0 + a;
        if (b == e) {                                            // link-less row: the unit self loop only
            if (tid == 0 && oe > ob) { oj[ob] = r; ox[ob] = 1.0f; }
            continue;
        }
        mark_columns(bitmap, W, w0, b, e, self_loop ? r : -1);
        const i32 per = (W + 255) / 256, wa = tid * per, wb = min(W, wa + per);
        u32 local = 0;
        for (i32 w = wa; w < wb; ++w) local += __popc(bitmap[w]);
        scratch[tid] = local;
        for (i32 k = ob + tid; k < oe; k += 256) oxu[k] = 0;
        __syncthreads();
        if (tid == 0) { u32 run = 0; for (int t = 0; t < 256; ++t) { const u32 v = scratch[t]; scratch[t] = run; run += v; } }
        __syncthreads();
        u32 run = scratch[tid];
        for (i32 w = wa; w < wb; ++w) { prefix[w] = run; run += __popc(bitmap[w]); }
        __syncthreads();
        for (i64 p = b + tid; p < e; p += 256) {
            const u32 c = (u32)(w0[p] & ID_MASK);
            const i32 k = ob + (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
            oj[k] = (i32)c;
            atomicAdd(&oxu[k], (u32)w1[p]);
        }
        __syncthreads();
        for (i32 k = ob + tid; k < oe; k += 256) ox[k] = (float)atomicOr(&oxu[k], 0u);   // read at L2; int -> float32, the cast of :368
        __syncthreads();
        if (self_loop && tid == 0) {
            const i32 k = ob + (i32)(prefix[r >> 5] + __popc(bitmap[r >> 5] & ((1u << (r & 31)) - 1u)));
            oj[k] = r;
            ox[k] = 1.0f;                                        // :362-364 (a flank key never has i == j)
        }
        __syncthreads();
    }
}

// ---- the same from SORTED RUNS: what an owner receives is one run per source rank, each already in row order (k_emit_by_row lays a
// chunk's entries out by matrix row) — so the rows need no partition at all, only the bounds of every row inside every run:
// a binary search per (row, run), done by the first lanes of the workgroup that owns the row.
constexpr int MAX_RUNS = 64;
__device__ __forceinline__ i64 run_lower_bound(const u64 *__restrict__ w0, i64 b, i64 e, u32 row) {
    while (b < e) {
        const i64 m = b + ((e - b) >> 1);
        if ((u32)(w0[m] >> ID_BITS) < row) b = m + 1; else e = m;
    }
    return b;
}
__device__ __forceinline__ i64 row_segments(i32 n_runs, const i64 *__restrict__ run_off, const u64 *__restrict__ w0, u32 row, i64 *seg_b, i64 *seg_e) {
    if ((int)threadIdx.x < n_runs) {
        const i64 b = run_off[threadIdx.x], e = run_off[threadIdx.x + 1];
        const i64 lo = run_lower_bound(w0, b, e, row);
        seg_b[threadIdx.x] = lo;
        seg_e[threadIdx.x] = run_lower_bound(w0, lo, e, row + 1);
    }
    __syncthreads();
    i64 total = 0;
    for (int s = 0; s < n_runs; ++s) total += seg_e[s] - seg_b[s];
    return total;
}
__device__ __forceinline__ void mark_columns_runs(u32 *bitmap, i32 W, const u64 *__restrict__ w0, i32 n_runs, const i64 *seg_b, const i64 *seg_e, i32 self_col) {
    for (i32 w = threadIdx.x; w < W; w += 256) bitmap[w] = 0;
    __syncthreads();
    for (int s = 0; s < n_runs; ++s)
        for (i64 p = seg_b[s] + threadIdx.x; p < seg_e[s]; p += 256) { const u32 c = (u32)(w0[p] & ID_MASK); atomicOr(&bitmap[c >> 5], 1u << (c & 31)); }
    if (self_col >= 0 && threadIdx.x == 0) atomicOr(&bitmap[self_col >> 5], 1u << (self_col & 31));
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_rows_distinct_runs(i32 n_local, i32 r0, i32 W, int self_loop, i32 n_runs, const i64 *__restrict__ run_off,
                                                            const u64 *__restrict__ w0, i32 *__restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    i64 *seg_b = (i64 *)smem, *seg_e = seg_b + MAX_RUNS;
    u32 *bitmap = (u32 *)(seg_e + MAX_RUNS);
    i32 *wsum = (i32 *)(bitmap + W);
    for (i32 a = blockIdx.x; a < n_local; a += gridDim.x) {
        const i64 total = row_segments(n_runs, run_off, w0, (u32)(r0 + a), seg_b, seg_e);
        if (total == 0) { if (threadIdx.x == 0) cnt[a] = self_loop ? 1 : 0; __syncthreads(); continue; }
        mark_columns_runs(bitmap, W, w0, n_runs, seg_b, seg_e, self_loop ? r0 + a : -1);
        i32 c = 0;
        for (i32 w = threadIdx.x; w < W; w += 256) c += __popc(bitmap[w]);
        c = wave_sum_i32(c);
        if (lane_id() == 0) wsum[threadIdx.x / HHX_WAVE] = c;
        __syncthreads();
        if (threadIdx.x == 0) cnt[a] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_rows_merge_emit_runs(i32 n_local, i32 r0, i32 W, int self_loop, i32 n_runs, const i64 *__restrict__ run_off,
                                                              const u64 *__restrict__ w0, const u64 *__restrict__ w1, const i32 *__restrict__ indptr,
                                                              i32 *__restrict__ oj, float *__restrict__ ox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    i64 *seg_b = (i64 *)smem, *seg_e = seg_b + MAX_RUNS;
    u32 *bitmap = (u32 *)(seg_e + MAX_RUNS), *prefix = bitmap + W, *scratch = prefix + W;
    const int tid = threadIdx.x;
    u32 *oxu = reinterpret_cast<u32 *>(ox);
    for (i32 a = blockIdx.x; a < n_local; a += gridDim.x) {
        const i32 ob = indptr[a], oe = indptr[a + 1], r = r0 + a;
        const i64 total = row_segments(n_runs, run_off, w0, (u32)r, seg_b, seg_e);
        if (total == 0) {                                        // link-less row: the unit self loop only
            if (tid == 0 && oe > ob) { oj[ob] = r; ox[ob] = 1.0f; }
            __syncthreads();
            continue;
        }
        mark_columns_runs(bitmap, W, w0, n_runs, seg_b, seg_e, self_loop ? r : -1);
        const i32 per = (W + 255) / 256, wa = tid * per, wb = min(W, wa + per);
        u32 local = 0;
        for (i32 w = wa; w < wb; ++w) local += __popc(bitmap[w]);
        scratch[tid] = local;
        for (i32 k = ob + tid; k < oe; k += 256) oxu[k] = 0;
        __syncthreads();
        if (tid == 0) { u32 run = 0; for (int t = 0; t < 256; ++t) { const u32 v = scratch[t]; scratch[t] = run; run += v; } }
        __syncthreads();
        u32 run = scratch[tid];
        for (i32 w = wa; w < wb; ++w) { prefix[w] = run; run += __popc(bitmap[w]); }
        __syncthreads();
        for (int sg = 0; sg < n_runs; ++sg)
            for (i64 p = seg_b[sg] + tid; p < seg_e[sg]; p += 256) {
                const u32 c = (u32)(w0[p] & ID_MASK);
                const i32 k = ob + (i32)(prefix[c >> 5] + __popc(bitmap[c >> 5] & ((1u << (c & 31)) - 1u)));
                oj[k] = (i32)c;
                atomicAdd(&oxu[k], (u32)w1[p]);
            }
        __syncthreads();
        for (i32 k = ob + tid; k < oe; k += 256) ox[k] = (float)atomicOr(&oxu[k], 0u);   // read at L2; int -> float32, the cast of :368
        __syncthreads();
        if (self_loop && tid == 0) {
            const i32 k = ob + (i32)(prefix[r >> 5] + __popc(bitmap[r >> 5] & ((1u << (r & 31)) - 1u)));
            oj[k] = r;
            ox[k] = 1.0f;                                        // :362-364 (a flank key never has i == j)
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int hhx_shard_create(hhx_ingest *h, const uint8_t *in_set_host, hhx_shard **out) {
    if (!h || !h->finalized || !in_set_host || !out) return fail("hhx_shard_create: bad argument (finalize the ingest first)");
    if (h->ord_limit > ((u64)1 << 31)) return fail("hhx_shard_create: more than 2^31 read pairs in the stream");
    const LinkRun *run = h->table(1);
    const i32 n_frag = h->t.n_frag;
    if (n_frag > (1 << 20)) return fail("hhx_shard_create: more than 2^20 fragments");
    auto *s = new hhx_shard();
    s->n_frag = n_frag;
    DevBuf<unsigned long long> first_pos;
    if (s->in_set.alloc((size_t)n_frag) || s->first.alloc((size_t)n_frag) || s->row_len.alloc((size_t)n_frag) || first_pos.alloc((size_t)n_frag)) { delete s; return 1; }
    int rc = 0;
    do {
        if (hipMemcpyAsync(s->in_set.p, in_set_host, (size_t)n_frag, hipMemcpyHostToDevice, g_stream) != hipSuccess) { rc = fail("hhx_shard_create: copy failed"); break; }
        int row_bits = 0;
        while (((i64)1 << row_bits) < n_frag) ++row_bits;
        if (run && run->n) {
            const SrcDirected src{run->key.p, run->ord_flank.p, run->fl.p, s->in_set.p};
            if ((rc = partition_records(src, DigRow(), 2 * run->n, row_bits, 9, &s->part, "d2m"))) break;
        }
        if (s->part.n_valid == 0) {
            (void)hipMemsetAsync(first_pos.p, 0xff, sizeof(unsigned long long) * (size_t)n_frag, g_stream);
            (void)hipMemsetAsync(s->row_len.p, 0, sizeof(i32) * (size_t)n_frag, g_stream);
        } else {
            k_row_first<<<grid_for((u64)n_frag * 64), 256, 0, g_stream>>>(n_frag, s->part.base.p, s->part.w1.p, first_pos.p, s->row_len.p);
        }
        k_first_signed<<<grid_for((u64)n_frag), 256, 0, g_stream>>>(n_frag, first_pos.p, s->first.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) { rc = fail("hhx_shard_create: kernel failed"); break; }
    } while (0);
    if (rc) { delete s; return rc; }
    *out = s;
    return 0;
}

extern "C" int hhx_shard_first(hhx_shard *s, void **first_dev) {
    if (!s || !first_dev) return fail("hhx_shard_first: null argument");
    *first_dev = s->first.p;
    return 0;
}

extern "C" int hhx_rank_first(i32 n_frag, const void *first_dev, void *frag_index_dev, i32 *n_linked) {
    if (n_frag <= 0 || !first_dev || !frag_index_dev) return fail("hhx_rank_first: bad argument");
    DevBuf<unsigned int> nl;
    if (nl.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(nl.p, 0, sizeof(unsigned int), g_stream));
    k_rank_first<<<(unsigned)((n_frag + 255) / 256), 256, 0, g_stream>>>(n_frag, (const unsigned long long *)first_dev, (i32 *)frag_index_dev, nl.p,
                                                                        (unsigned long long)SHARD_NONE);
    HHX_LAUNCH_CHECK();
    unsigned int v = 0;
    HHX_HIP(hipMemcpyAsync(&v, nl.p, sizeof v, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (n_linked) *n_linked = (i32)v;
    return 0;
}

extern "C" int hhx_shard_emit(hhx_shard *s, const void *frag_index_dev, i32 n_bounds, const i32 *bounds, void **w0_dev, void **w1_dev, i64 *counts) {
    if (!s || !frag_index_dev || n_bounds < 2 || !bounds || !w0_dev || !w1_dev || !counts) return fail("hhx_shard_emit: bad argument");
    const i32 shape = bounds[n_bounds - 1];
    const i64 n = s->part.n_valid;
    DevBuf<i64> len_by_row, row_off;
    if (len_by_row.alloc((size_t)shape + 1) || row_off.alloc((size_t)shape + 2)) return 1;
    HHX_HIP(hipMemsetAsync(len_by_row.p, 0, sizeof(i64) * ((size_t)shape + 1), g_stream));
    k_len_by_row<<<grid_for((u64)s->n_frag), 256, 0, g_stream>>>(s->n_frag, (const i32 *)frag_index_dev, s->row_len.p, len_by_row.p);
    HHX_LAUNCH_CHECK();
    i64 total = 0;
    HHX_TRY(exclusive_scan_i64(len_by_row.p, row_off.p, shape, &total));
    if (total != n) return fail("hhx_shard_emit: %lld entries but the ranked fragments hold %lld (fragment index does not cover this chunk)", (long long)n, (long long)total);
    if (s->out_w0.alloc((size_t)n + 1) || s->out_w1.alloc((size_t)n + 1)) return 1;
    if (n) {
        k_emit_by_row<<<grid_for((u64)s->n_frag * 64), 256, 0, g_stream>>>(s->n_frag, s->part.base.p, s->part.w0.p, s->part.w1.p, (const i32 *)frag_index_dev,
                                                                           row_off.p, s->out_w0.p, s->out_w1.p);
        HHX_LAUNCH_CHECK();
    }
    std::vector<i64> at((size_t)n_bounds);
    for (i32 k = 0; k < n_bounds; ++k) {
        if (bounds[k] < 0 || bounds[k] > shape || (k && bounds[k] < bounds[k - 1])) return fail("hhx_shard_emit: bad row bounds");
        HHX_HIP(hipMemcpyAsync(&at[(size_t)k], row_off.p + bounds[k], sizeof(i64), hipMemcpyDeviceToHost, g_stream));
    }
    HHX_HIP(hipStreamSynchronize(g_stream));
    for (i32 k = 0; k + 1 < n_bounds; ++k) counts[k] = at[(size_t)k + 1] - at[(size_t)k];
    *w0_dev = s->out_w0.p;
    *w1_dev = s->out_w1.p;
    return 0;
}

extern "C" int hhx_shard_destroy(hhx_shard *s) {
    delete s;
    return 0;
}

extern "C" int hhx_rows_from_entries(i64 n, const void *w0_dev, const void *w1_dev, i32 r0, i32 r1, i32 shape, int add_self_loops, hhx_csr **out) {
    if (n < 0 || r0 < 0 || r1 < r0 || r1 > shape || !out || (n && (!w0_dev || !w1_dev))) return fail("hhx_rows_from_entries: bad argument");
    const i32 n_local = r1 - r0;
    const i32 W = (shape + 31) / 32;
    const size_t lds = (size_t)W * 8 + 256 * 4;
    if (lds > 160 * 1024) return fail("hhx_rows_from_entries: order %d exceeds the LDS bitmap capacity", shape);
    Partitioned<u64> part;
    if (n && n_local) {
        int row_bits = 0;
        while (((i64)1 << row_bits) < n_local) ++row_bits;
        const SrcRecs<u64> src{(const u64 *)w0_dev, (const u64 *)w1_dev};
        HHX_TRY(partition_records(src, DigLocalRow{(u32)r0}, n, row_bits, 9, &part, "rows"));
    } else if (n) {
        return fail("hhx_rows_from_entries: entries for an empty row block");
    }
    const bool any = part.n_valid > 0;
    DevBuf<i32> cnt, indptr;
    if (cnt.alloc((size_t)n_local + 1) || indptr.alloc((size_t)n_local + 2)) return 1;
    static int attr_set = -1;           // the attribute is per device: keyed on the current ordinal
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_rows_distinct, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_rows_merge_emit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = attr_dev;
    }
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(n_local, 256 * 8));
    if (n_local) {
        k_rows_distinct<<<grid, 256, (size_t)W * 4 + 16, g_stream>>>(n_local, r0, W, add_self_loops, any ? part.base.p : nullptr, any ? part.w0.p : nullptr, cnt.p);
        HHX_LAUNCH_CHECK();
    }
    i64 nnz = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, indptr.p, n_local, &nnz));
    hhx_csr *m = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(n_local, shape, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->indptr.p, indptr.p, sizeof(i32) * ((size_t)n_local + 1), hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess && n_local)
        k_rows_merge_emit<<<grid, 256, lds, g_stream>>>(n_local, r0, W, add_self_loops, any ? part.base.p : nullptr, any ? part.w0.p : nullptr,
                                                        any ? part.w1.p : nullptr, indptr.p, m->indices.p, m->data.p);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("hhx_rows_from_entries: %s", hipGetErrorString(e)); }
    *out = m;
    return 0;
}

// The same row block from what the all-to-all(v) delivers as it is: n_runs runs (one per source rank, run k = entries
// [run_off[k], run_off[k + 1])), each sorted by row because hhx_shard_emit writes a chunk's entries in matrix-row order.  No
// partition pass: 19.7 -> (measured below) ms at n = 100k / 330 M entries.
extern "C" int hhx_rows_from_runs(i32 n_runs, const i64 *run_off_host, const void *w0_dev, const void *w1_dev, i32 r0, i32 r1, i32 shape,
                                  int add_self_loops, hhx_csr **out) {
    if (n_runs < 1 || n_runs > MAX_RUNS || !run_off_host || r0 < 0 || r1 < r0 || r1 > shape || !out) return fail("hhx_rows_from_runs: bad argument");
    const i64 n = run_off_host[n_runs];
    if (n < 0 || (n && (!w0_dev || !w1_dev))) return fail("hhx_rows_from_runs: bad argument");
    const i32 n_local = r1 - r0;
    if (n && !n_local) return fail("hhx_rows_from_runs: entries for an empty row block");
    const i32 W = (shape + 31) / 32;
    const size_t lds_fixed = (size_t)MAX_RUNS * 16, lds = lds_fixed + (size_t)W * 8 + 256 * 4;
    if (lds > 160 * 1024) return fail("hhx_rows_from_runs: order %d exceeds the LDS bitmap capacity", shape);
    DevBuf<i64> run_off;
    DevBuf<i32> cnt, indptr;
    if (run_off.alloc((size_t)n_runs + 1) || cnt.alloc((size_t)n_local + 1) || indptr.alloc((size_t)n_local + 2)) return 1;
    HHX_HIP(hipMemcpyAsync(run_off.p, run_off_host, sizeof(i64) * ((size_t)n_runs + 1), hipMemcpyHostToDevice, g_stream));
    static int attr_set = -1;
    int attr_dev = 0;
    HHX_HIP(hipGetDevice(&attr_dev));
    if (attr_set != attr_dev) {
        HHX_HIP(hipFuncSetAttribute((const void *)k_rows_distinct_runs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HHX_HIP(hipFuncSetAttribute((const void *)k_rows_merge_emit_runs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = attr_dev;
    }
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(n_local, 256 * 8));
    if (n_local) {
        KTimer kt("rows_distinct");
        k_rows_distinct_runs<<<grid, 256, lds_fixed + (size_t)W * 4 + 16, g_stream>>>(n_local, r0, W, add_self_loops, n_runs, run_off.p, (const u64 *)w0_dev, cnt.p);
        HHX_LAUNCH_CHECK();
    }
    i64 nnz = 0;
    HHX_TRY(exclusive_scan_i32(cnt.p, indptr.p, n_local, &nnz));
    hhx_csr *m = nullptr;
    HHX_TRY(hhx_csr_alloc_internal(n_local, shape, nnz, &m));
    hipError_t e = hipMemcpyAsync(m->indptr.p, indptr.p, sizeof(i32) * ((size_t)n_local + 1), hipMemcpyDeviceToDevice, g_stream);
    if (e == hipSuccess && n_local) {
        KTimer kt("rows_emit");
        k_rows_merge_emit_runs<<<grid, 256, lds, g_stream>>>(n_local, r0, W, add_self_loops, n_runs, run_off.p, (const u64 *)w0_dev, (const u64 *)w1_dev, indptr.p,
                                                             m->indices.p, m->data.p);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) { hhx_csr_free(m); return fail("hhx_rows_from_runs: %s", hipGetErrorString(e)); }
    *out = m;
    return 0;
}
