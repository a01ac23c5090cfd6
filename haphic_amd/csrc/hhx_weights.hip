// Link weights on the flank table: the three in-place dict rewrites of the reference that sit
// between the ingest and dict_to_matrix, as one gather kernel over the (frag_i, frag_j, value) arrays in dict order.
//   normalize_by_nlinks   :718-724   value /= (links[i] * links[j]) ** 0.5         (--normalize_by_nlinks :2895)
//   normalize_by_length   :727-738   value /= (fl_i / 1e6) * (fl_j / 1e6), fl = min(len, 2 * flank bp)   (never called by the
//                                    reference — dead code — kept for completeness)
//   reduce_inter_hap_HiC_links :695-707   value -= value * phasing_weight when the two fragments carry different haplotype
//                                    tags; entries that reach 0 are deleted from the dict (the caller compacts: n_zero > 0)
// Values are Python floats (float64) in the reference and are cast to float32 only when the matrix is built (:368);
// the same here: the kernel works in float64 on the value array that hhx_dict_to_matrix / hhx_ingest_flank_device use.
// HBM-bound: 16 B read + 8 B written per key, two 8-byte gathers from a per-fragment table that lives in L2.
#include "hhx_common.h"

using namespace hhx;

namespace {

// MODE 0: per-fragment table = int64 link totals; 1: int64 fragment lengths (param = 2 * flank in bp); 2: int32 haplotype tags
template <int MODE>
__global__ __launch_bounds__(256) void k_link_weights(i64 n, const i32 *__restrict__ fi, const i32 *__restrict__ fj, double *__restrict__ value,
                                                      const i64 *__restrict__ per_frag, const i32 *__restrict__ tag, double param,
                                                      unsigned long long *__restrict__ n_zero) {
    i64 zeros = 0;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const i32 a = fi[k], b = fj[k];
        double v = value[k];
        if (MODE == 0) {
            // Python: (int * int) ** 0.5 — the product is exact (< 2^53 for any real link totals); float.__pow__ is C pow(),
            // which glibc rounds correctly in all but a handful of cases, i.e. to the IEEE square root: sqrt() here (the
            // device pow() is 1-2 ulp off, measured against the reference's values)
            // each factor is converted before the multiplication: an int64 product would wrap beyond ~3e9 links per fragment, where
            // Python's integers stay exact (the double product is then rounded once, like float(int * int))
            v /= sqrt((double)per_frag[a] * (double)per_frag[b]);
        } else if (MODE == 1) {
            const double two_flanks = param;
            const double la = (double)per_frag[a], lb = (double)per_frag[b];
            const double fa = la <= two_flanks ? la : two_flanks, fb = lb <= two_flanks ? lb : two_flanks;
            v /= (fa / 1000000.0) * (fb / 1000000.0);
        } else {
            if (tag[a] != tag[b]) {
                v -= v * param;
                if (v == 0.0) ++zeros;
            }
        }
        value[k] = v;
    }
    if (MODE == 2) {
        zeros = wave_sum_i64(zeros);
        if (lane_id() == 0 && zeros) atomicAdd(n_zero, (unsigned long long)zeros);
    }
}

}  // namespace

extern "C" int hhx_link_weights(i64 n_keys, const i32 *frag_i, const i32 *frag_j, double *value, int on_device, int mode, i32 n_frag,
                                const i64 *per_frag_host, const i32 *tag_host, double param, i64 *n_zero) {
    if (n_zero) *n_zero = 0;
    if (n_keys < 0 || n_frag < 0 || mode < 0 || mode > 2) return fail("hhx_link_weights: bad arguments");
    if (n_keys == 0) return 0;
    if (!frag_i || !frag_j || !value) return fail("null pointer");
    if ((mode == 2 && !tag_host) || (mode != 2 && !per_frag_host)) return fail("hhx_link_weights: the per-fragment table of this mode is missing");
    DevBuf<i32> di, dj, dtag;
    DevBuf<double> dv;
    DevBuf<i64> dper;
    DevBuf<unsigned long long> dz;
    const i32 *pi = frag_i, *pj = frag_j;
    double *pv = value;
    if (!on_device) {
        if (di.alloc((size_t)n_keys) || dj.alloc((size_t)n_keys) || dv.alloc((size_t)n_keys)) return 1;
        HHX_HIP(hipMemcpyAsync(di.p, frag_i, sizeof(i32) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(dj.p, frag_j, sizeof(i32) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(dv.p, value, sizeof(double) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
        pi = di.p; pj = dj.p; pv = dv.p;
    }
    if (dz.alloc(1)) return 1;
    HHX_HIP(hipMemsetAsync(dz.p, 0, sizeof(unsigned long long), g_stream));
    if (mode == 2) {
        if (dtag.alloc((size_t)n_frag + 1)) return 1;
        HHX_HIP(hipMemcpyAsync(dtag.p, tag_host, sizeof(i32) * (size_t)n_frag, hipMemcpyHostToDevice, g_stream));
    } else {
        if (dper.alloc((size_t)n_frag + 1)) return 1;
        HHX_HIP(hipMemcpyAsync(dper.p, per_frag_host, sizeof(i64) * (size_t)n_frag, hipMemcpyHostToDevice, g_stream));
    }
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>((n_keys + 255) / 256, 256 * 16));
    {
        KTimer kt("link_weights");
        if (mode == 0) k_link_weights<0><<<grid, 256, 0, g_stream>>>(n_keys, pi, pj, pv, dper.p, nullptr, param, dz.p);
        else if (mode == 1) k_link_weights<1><<<grid, 256, 0, g_stream>>>(n_keys, pi, pj, pv, dper.p, nullptr, param, dz.p);
        else k_link_weights<2><<<grid, 256, 0, g_stream>>>(n_keys, pi, pj, pv, nullptr, dtag.p, param, dz.p);
    }
    HHX_LAUNCH_CHECK();
    unsigned long long z = 0;
    HHX_HIP(hipMemcpyAsync(&z, dz.p, sizeof z, hipMemcpyDeviceToHost, g_stream));
    if (!on_device) HHX_HIP(hipMemcpyAsync(value, dv.p, sizeof(double) * (size_t)n_keys, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    if (n_zero) *n_zero = (i64)z;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// f3: reassign's link-density input, HapHiC_reassign.py parse_link_dict :217-263 (normalize_by_nlinks = False branch):
//     ctg_group_link_dict[ctg][group] = sum of the links between ctg and the contigs of `group` (groups != 'ungrouped')
// for every contig pair of full_link_dict — a sparse matrix times the n_ctg x n_groups group indicator.  One thread per
// key: two 64-bit integer atomic adds into the dense n_ctg x n_groups table (order-free, exact: the links are integer
// counts) and two atomic mins recording the dict position of the first contribution — the inner dicts of the reference
// are insertion ordered, and run_reassignment iterates them.
namespace {
__global__ __launch_bounds__(256) void k_group_link_sums(i64 n, const i32 *__restrict__ fi, const i32 *__restrict__ fj, const i64 *__restrict__ links,
                                                         const i32 *__restrict__ group, i32 n_groups, unsigned long long *__restrict__ sums,
                                                         unsigned long long *__restrict__ first) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (i64)gridDim.x * blockDim.x) {
        const i32 a = fi[k], b = fj[k];
        const i32 ga = group[a], gb = group[b];
        const unsigned long long v = (unsigned long long)links[k];
        if (gb >= 0) {                                   // add_ctg_group(ctg_i, group_j, links) :246
            atomicAdd(&sums[(size_t)a * n_groups + gb], v);
            atomicMin(&first[(size_t)a * n_groups + gb], (unsigned long long)(2 * k));
        }
        if (ga >= 0) {                                   // add_ctg_group(ctg_j, group_i, links) :247
            atomicAdd(&sums[(size_t)b * n_groups + ga], v);
            atomicMin(&first[(size_t)b * n_groups + ga], (unsigned long long)(2 * k + 1));
        }
    }
}
}  // namespace

extern "C" int hhx_group_link_sums(i64 n_keys, const i32 *frag_i, const i32 *frag_j, const i64 *links, i32 n_ctg, const i32 *group_host,
                                   i32 n_groups, i64 *sums_host, i64 *first_host) {
    if (n_keys < 0 || n_ctg < 0 || n_groups < 0) return fail("hhx_group_link_sums: bad arguments");
    const size_t cells = (size_t)n_ctg * (size_t)n_groups;
    if (cells == 0) return 0;
    if (!sums_host || !first_host || !group_host || (n_keys && (!frag_i || !frag_j || !links))) return fail("null pointer");
    DevBuf<i32> di, dj, dg;
    DevBuf<i64> dl;
    DevBuf<unsigned long long> ds, df;
    if (di.alloc((size_t)n_keys + 1) || dj.alloc((size_t)n_keys + 1) || dl.alloc((size_t)n_keys + 1) || dg.alloc((size_t)n_ctg) || ds.alloc(cells) ||
        df.alloc(cells)) return 1;
    if (n_keys) {
        HHX_HIP(hipMemcpyAsync(di.p, frag_i, sizeof(i32) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(dj.p, frag_j, sizeof(i32) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
        HHX_HIP(hipMemcpyAsync(dl.p, links, sizeof(i64) * (size_t)n_keys, hipMemcpyHostToDevice, g_stream));
    }
    HHX_HIP(hipMemcpyAsync(dg.p, group_host, sizeof(i32) * (size_t)n_ctg, hipMemcpyHostToDevice, g_stream));
    HHX_HIP(hipMemsetAsync(ds.p, 0, sizeof(unsigned long long) * cells, g_stream));
    HHX_HIP(hipMemsetAsync(df.p, 0xff, sizeof(unsigned long long) * cells, g_stream));
    if (n_keys) {
        KTimer kt("group_link_sums");
        k_group_link_sums<<<(unsigned)std::max<i64>(1, std::min<i64>((n_keys + 255) / 256, 256 * 16)), 256, 0, g_stream>>>(
            n_keys, di.p, dj.p, dl.p, dg.p, n_groups, ds.p, df.p);
        HHX_LAUNCH_CHECK();
    }
    HHX_HIP(hipMemcpyAsync(sums_host, ds.p, sizeof(i64) * cells, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipMemcpyAsync(first_host, df.p, sizeof(i64) * cells, hipMemcpyDeviceToHost, g_stream));
    HHX_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
