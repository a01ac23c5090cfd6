/*
 * haphic_hip.h — C ABI of libhaphic_hip.so: MI355X (gfx950) implementation of HapHiC's Hi-C
 * link-matrix construction + Markov clustering hot path (scripts/HapHiC_cluster.py).
 *
 * The reference has no FFI/plugin API; its narrowest seams are module-level Python functions
 * (SURVEY.md §8b, S1–S6).  Every entry point below names the reference interface it replaces
 * (file:line, all in scripts/HapHiC_cluster.py).  The ctypes binding a maintainer would add is
 * haphic_amd/_lib.py; the re-binding of the reference's names is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; hhx_last_error() gives the
 *     thread-local message (the Python shim raises RuntimeError, as the reference does, e.g. :2507).
 *   - plain pointers and sizes only.  "host" pointers are ordinary memory; "dev" pointers are HIP
 *     device memory on the current device.  Kernels run on the stream set by hhx_set_stream()
 *     (default: the null stream).  ONE stream at a time per process: the library's memory pool recycles blocks in
 *     stream order; hhx_set_stream drains the device when the stream changes, and concurrent callers on different
 *     streams are not supported.
 *   - matrices: the reference's column-stochastic M is scipy CSC (indptr,indices,data) with int32
 *     indices and float32 data.  That triple is byte-for-byte CSR of T = M^T; the library works on
 *     CSR(T).  "row" below == "column" in the reference.  Rows are kept sorted by index.
 *   - hhx_csr is an opaque device-resident matrix owned by the library (release with hhx_csr_free).
 */
#ifndef HAPHIC_HIP_H
#define HAPHIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hhx_csr hhx_csr;
typedef struct hhx_ingest hhx_ingest;

/* ---------------------------------------------------------------- runtime */
const char *hhx_last_error(void);
int hhx_version(void);
int hhx_device_count(int *count);
int hhx_set_device(int device);
int hhx_set_stream(void *hip_stream);          /* hipStream_t; NULL = null stream (thread-local) */
int hhx_synchronize(void);
int hhx_pool_trim(void);                       /* release cached device memory (a block of hhx_pool_prewarm nobody has used yet survives ONE such call) */
/* the same, but the blocks of at most 64 MiB and, largest first, up to keep_bytes of the blocks of at most 8 GiB stay cached: between two steps of one job the
 * transient blocks of the first (tens of GB at C3) go back to the driver while the second still finds blocks for its own mid-size buffers — a fresh block costs
 * 12-30 ms per GB, and 0.1-0.6 s a call while another thread of the process is releasing memory (measured inside the whole C3 run: DESIGN.md 1.2) */
int hhx_pool_trim_keep(int64_t keep_bytes);
/* take blocks of these sizes from the driver now and leave them in the pool's cache (thread-safe; meant for a helper thread of the caller while a long
 * kernel runs: the next step's pools then cost no fresh device memory).  Stops quietly when the device has no room.  Such a block survives the next
 * hhx_pool_trim (not the one after) unless it has been handed out in between; when the device runs out of memory it is released like every other cached block. */
int hhx_pool_prewarm(int32_t n, const int64_t *bytes);
/* per-kernel device timing with HIP events on the launch stream (bench.py's roofline leg):
 * names: "ingest" (map + partition + aggregate of one push), "aggregate", "ingest_merge", "link_matrix",
 * "spgemm_symbolic", "spgemm_numeric", "expand_window", "expand_window_short", "expand_hash", "expand_compact",
 * "expand_tiny", "inflate_stats", "prune_write", "convergence", "class_layout", "d2m_*" (the stages of the link-matrix
 * build), "bin_contacts" */
/* Which kernel class / arithmetic / layout the calls take (no reference counterpart).  EVERY setting of every knob gives the same
 * results — the verification tests switch classes with them and compare bits; bench.py / tools time the variants.  Known names
 * (anything else is refused): "cls" 0 = stream iteration 0 as (column, value) pairs instead of the class stream; "cls_nc",
 * "cls_balance" (layout of the class stream); "links_integer" 0 = float arithmetic for iteration 0 (results within float32
 * round-off, not the same bits: the other specification of DESIGN.md 2); "links_sym" 0 = every row walks all its products;
 * "dense_tri" 1 / 0 = force / forbid the upper-block-triangle storage of the dense block; "hash_max" = largest product count of a
 * row that tries the LDS hash table first (0: window / compact classes only); "tile_u", "win_batch", "cache_slice_mb" (tile
 * shapes and window count of the window kernel); "block_tiles" 0 = iterations >= 1 of the window class walk the stream in tiles
 * of one (B row, window) segment each instead of 64-entry blocks (default 15; 1, 2, 5, 11, 31: other block-tile shapes /
 * addressings); "row_order" 0 = the window-class rows as listed instead of in min-hash order; "reuse" 4 / 2 = the grouped kernel;
 * "dense_seed_hint" (the sweep's first pool sizes).  value INT64_MIN: back to the default.  Unset knobs fall back to the
 * environment variable HHX_<NAME>. */
int hhx_tune(const char *name, int64_t value);
int hhx_profile_enable(int on);
int hhx_profile_reset(void);
int hhx_profile_get(const char *kernel, double *total_ms, int64_t *launches);
/* event counters gathered while profiling is on: "expand_window_products" (products streamed by
 * k_expand_window_pass), "expand_window_a_reads" (entries of A staged, summed over the column windows),
 * "ingest_records" (read pairs that survive the map stage) */
int hhx_profile_counter(const char *name, int64_t *value);

/* ---------------------------------------------------------------- matrices */
/* build from / copy to host arrays (numpy: csc.indptr, csc.indices, csc.data) */
int hhx_csr_from_host(int32_t n_rows, int32_t n_cols, const int32_t *indptr, const int32_t *indices,
                      const float *data, hhx_csr **out);
/* build by copying device arrays (e.g. torch tensors after an all-gather) */
int hhx_csr_from_device(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *dev_indptr,
                        const int32_t *dev_indices, const float *dev_data, hhx_csr **out);
int hhx_csr_shape(const hhx_csr *m, int32_t *n_rows, int32_t *n_cols, int64_t *nnz);
int hhx_csr_to_host(const hhx_csr *m, int32_t *indptr, int32_t *indices, float *data);
int hhx_csr_device_ptrs(const hhx_csr *m, void **dev_indptr, void **dev_indices, void **dev_data);
int hhx_csr_copy(const hhx_csr *m, hhx_csr **out);
/* rows [r0, r1) as a new (r1-r0) x n_cols matrix: the row-block shard of SURVEY §8e */
int hhx_csr_row_block(const hhx_csr *m, int32_t r0, int32_t r1, hhx_csr **out);
/* row blocks stacked in order (same column count; fewer than 2^31 entries in total) */
int hhx_csr_vstack(int32_t n_blocks, const hhx_csr *const *blocks, hhx_csr **out);
/* SURVEY §8e, the per-iteration all-gather(v) of the pruned row blocks (mcl :2026-2062 with T sharded by row block): RCCL has no
 * all-gather-v, so a block travels as ONE int32 message [row lengths | column indices | float32 value bits] padded to the longest
 * message of the world.  hhx_csr_pack_block writes m's message (n_rows + 2 nnz words) into a caller-owned device buffer;
 * hhx_csr_unpack_blocks turns the gathered messages (message b at packed + b * stride_words, rows[b] rows / nnz[b] entries — host
 * arrays, from the header exchange) into the stacked matrix: three device copies per block and one scan for the row pointer. */
int hhx_csr_pack_block(const hhx_csr *m, void *dst_dev, int64_t capacity_words);
int hhx_csr_unpack_blocks(int32_t n_blocks, const int64_t *rows, const int64_t *nnz, const void *packed_dev, int64_t stride_words,
                          int32_t n_cols, hhx_csr **out);
/* free / total device memory in bytes (memory cached by the library's pool counts as used: hhx_pool_trim first) */
int hhx_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* bytes cached by the library's pool: reusable by the library's next allocations without asking the driver */
int hhx_pool_cached_bytes(int64_t *bytes);
int hhx_csr_free(hhx_csr *m);

/* ---------------------------------------------------------------- S3: normalize / power / prune
 * sklearn.preprocessing.normalize(M, norm='l1', axis=0), call sites :2014 :2038 :2144 — in place. */
int hhx_normalize_l1(hhx_csr *m);
/* `normalize(matrix.power(inflation), norm='l1', axis=0)` :2037-2038 — in place. */
int hhx_inflate(hhx_csr *m, double inflation);
/* prune(matrix, pruning, dense_matrix=False) :1987-2014: keep entries >= pruning, restore the row
 * maximum, L1-normalise. */
int hhx_prune(const hhx_csr *m, double pruning, hhx_csr **out);
/* :2037-2042 fused: prune(normalize(power(C, inflation))); C is consumed (its data is overwritten). */
int hhx_inflate_prune(hhx_csr *c, double inflation, double pruning, hhx_csr **out);
/* same result, c untouched: for a row block of the pre-expanded matrix that run_mcl_clustering's inflation sweep
 * (:2155-2158) revisits for every inflation */
int hhx_inflate_prune_keep(const hhx_csr *c, double inflation, double pruning, hhx_csr **out);

/* ---------------------------------------------------------------- S1: expansion
 * sparse_dot_mkl.dot_product_mkl(A_csc, B_csc) :39-43, used by mkl_matrix_power :2017-2023.
 * On the CSC view out = A_csc * B_csc; in CSR(T) terms the caller passes (a = T_B, b = T_A), i.e.
 * out_T = T_B * T_A.  a may be a row block (n_local x k) of the left operand (multi-GPU shard).
 * Accumulation: each float32 product is formed exactly in double, rounded to the nearest multiple of
 * 2^-fx_shift (ties to even; shift chosen from ||a||_inf * max|b|) and summed exactly (64-bit integer adds),
 * so the result is independent of summation order and of the GPU count; it is rounded to float32 once. */
int hhx_spgemm(const hhx_csr *a, const hhx_csr *b, hhx_csr **out);
/* same with an explicit fixed-point shift (tests); products counted into *n_products if non-NULL */
int hhx_spgemm_ex(const hhx_csr *a, const hhx_csr *b, int fx_shift, hhx_csr **out, int64_t *n_products);

/* One fused MCL iteration on the device, :2030-2042: out = prune(normalize(power(a * b, inflation))),
 * the expanded matrix a*b is consumed row by row in LDS and never written to HBM (at n = 100k the
 * un-pruned pre-expansion of :2147 has 10^10 entries).  a may be a row block.  Operands must be
 * stochastic (entries in [0,1], row sums <= 1): products are rounded on the 2^-fx_shift grid, fx_shift <= 52
 * (default 52), and accumulated with exact float64 adds. */
int hhx_expand_inflate_prune(const hhx_csr *a, const hhx_csr *b, int fx_shift, double inflation, double pruning,
                             hhx_csr **out, int64_t *n_products, int64_t *nnz_expanded);

/* ---------------------------------------------------------------- S2: convergence + whole mcl()
 * :2044-2046  d = abs(M - last) - 1e-5*abs(last); *stat = max(0, d.max()) evaluated in float32. */
int hhx_convergence_stat(const hhx_csr *m, const hhx_csr *last, float *stat);
/* mcl(matrix, expansion, inflation, iters, pruning, dense_matrix=False) :2026-2062 on the
 * pre-expanded matrix (run_mcl_clustering :2144-2147).  Not converging is not an error (:2058-2062):
 * *converged = 0 and the last matrix is returned.  stats (may be NULL) receives 4 int64 per executed
 * iteration: nnz entering, nnz after expansion, nnz after pruning, number of products. */
int hhx_mcl(const hhx_csr *pre_expanded, int expansion, double inflation, int max_iter, double pruning,
            hhx_csr **out, int *n_iter, int *converged, int64_t *stats);
/* run_mcl_clustering :2144-2158 for one inflation, starting from the L1-normalised link matrix: the
 * pre-expansion (:2146-2147) is fused into iteration 0 instead of being materialised. */
/* mcl() :2026-2062 picked up after its first `done` (>= 1) iterations, `m` being the matrix they left; n_iter and
 * stats[] keep counting from `done`.  run_mcl_clustering at orders where M^e has more than 2^31 entries keeps M^e in
 * HBM as row blocks (hhx_spgemm on hhx_csr_row_block), forms iteration 0 of every inflation with
 * hhx_inflate_prune_keep per block + hhx_csr_vstack, and resumes here: the expansion of the link matrix (a second
 * per inflation at n = 100k) is paid once for the whole sweep, as in the reference (:2146-2147). */
int hhx_mcl_resume(const hhx_csr *m, int done, int expansion, double inflation, int max_iter, double pruning,
                   hhx_csr **out, int *n_iter, int *converged, int64_t *stats);
int hhx_mcl_normalized(const hhx_csr *normalized, int expansion, double inflation, int max_iter, double pruning,
                       hhx_csr **out, int *n_iter, int *converged, int64_t *stats);
/* The inflation sweep of run_mcl_clustering :2155-2158 (every inflation restarts from the matrix pre-expanded at :2146-2147) with ONE
 * expansion.  hhx_expand_links_dense: rows [r0, r1) of M^2 — M the L1-normalised (:2144) raw link matrix of dict_to_matrix — as a
 * dense float32 row block in HBM (4 B x (r1 - r0) x n; all rows of a symmetric integer matrix whose square does not fit: the upper block
 * triangle alone, 0.55 x 4 B x n^2; hhx_dense_shape reports the bytes).  hhx_dense_inflate_prune: iteration 0 of
 * mcl() (:2037-2042: power, normalise, prune, restore the maximum, normalise) of those rows at one inflation, bit for bit what
 * hhx_mcl_links computes in its first iteration; stack the blocks (hhx_csr_vstack) and continue with hhx_mcl_resume(done = 1). */
typedef struct hhx_dense hhx_dense;
/* upper_only = 1 (integer arithmetic only): fill just the blocks (I, J >= I) of the rows — 60 % of their products — and leave the
 * columns left of a row's own block to the caller, who owns the mirror image: the ranks of the multi-GPU driver exchange them
 * (hhx_dense_device gives the block and its column-window plan).  hhx_links_integer_ok says whether the arithmetic applies. */
int hhx_expand_links_dense(const hhx_csr *links, int32_t r0, int32_t r1, int fx_shift, int upper_only, hhx_dense **out,
                           int64_t *n_products, int64_t *nnz_expanded);
int hhx_links_integer_ok(const hhx_csr *links, int *ok, int *shift);
/* what iteration 0 of hhx_mcl_links will do with this matrix on this device: *integer (the arithmetic above applies); *layout = 1 the
 * symmetric half into the square float32 block of all rows (+ transposition), 2 into its upper block triangle alone (the square
 * would take too large a share of the device), 0 every row walks all its products into the fused epilogue */
int hhx_links_plan(const hhx_csr *links, int *integer, int *layout);
int hhx_dense_device(const hhx_dense *d, void **x_dev, int64_t *ld, int32_t *cap_win, int32_t *n_win);   /* ld: row pitch in floats (>= n_cols) */
/* rectangles of such a block on their way between ranks (device pointers, sizes and pitches in floats): hhx_copy_rect_f32 packs /
 * places a rows x cols rectangle, hhx_transpose_f32 writes its transpose (dst is cols x rows) — the mirror image of the symmetric
 * pre-expansion, which a rank receives as another rank's rows */
int hhx_copy_rect_f32(const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols);
int hhx_transpose_f32(const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols);
int hhx_dense_inflate_prune(const hhx_dense *d, double inflation, double pruning, hhx_csr **out);
/* the same for k <= 8 inflations in ONE pass over the block (x = y / d_i and log2(x) are formed once per entry, the 4 B x n^2 block
 * is read once): outs[i] is bit for bit what hhx_dense_inflate_prune(d, inflations[i], ...) returns */
int hhx_dense_inflate_prune_multi(const hhx_dense *d, int k, const double *inflations, double pruning, hhx_csr **outs);
int hhx_dense_shape(const hhx_dense *d, int32_t *n_rows, int32_t *n_cols, int64_t *bytes);
int hhx_dense_free(hhx_dense *d);

/* run_mcl_clustering :2144-2158 for one inflation straight from the RAW link matrix that dict_to_matrix
 * returns (:362-368): the L1 normalisation (:2144), the pre-expansion (:2146-2147, fused into iteration 0) and
 * mcl().  When the matrix holds integer link counts <= 65535 (always, unless --normalize_by_nlinks / GFA weights
 * were applied) iteration 0 streams its right operand as the CLASS STREAM: the count-1 entries of every (row, column
 * window) segment — 75 % of a Hi-C link matrix — go as 16-bit columns alone, 2 B per product instead of 6.
 * When it is also SYMMETRIC with row sums below 2^18 (every matrix of dict_to_matrix at the sizes of BASELINE.json) the
 * pre-expansion runs in INTEGER arithmetic, (M^2)_ij = S_ij / d_i with S = L D^-1 L evaluated exactly (DESIGN.md
 * 4.1; within 3e-7 of the float32-normalised product, far inside the float32 accumulation noise of the
 * reference's own SpGEMM): S is symmetric bit for bit, so only its upper block triangle is computed and the rest is
 * transposed — 60 % of the products at five column windows.  Other matrices take the float arithmetic of
 * hhx_normalize_l1 + hhx_mcl_normalized (bit-identical to it). */
int hhx_mcl_links(const hhx_csr *links, int expansion, double inflation, int max_iter, double pruning,
                  hhx_csr **out, int *n_iter, int *converged, int64_t *stats);
/* products_host[i] = sum over the entries (i, k) of a of nnz(row k of b): the cost of row i of a * b.  The multi-GPU driver
 * cuts the row blocks of the expansion at equal product counts with it (SURVEY §8e "row-block ... by balanced nnz"). */
int hhx_row_products(const hhx_csr *a, const hhx_csr *b, int64_t *products_host);
/* Iteration 0 of hhx_mcl_links for ONE ROW BLOCK (multi-GPU shard): rows [r0, r1) of the result, links = the WHOLE raw
 * link matrix (all-gathered).  Same arithmetic, hence the same bits, as the one-GPU call (a row block cannot use the symmetry:
 * it walks all its products). */
int hhx_expand_links(const hhx_csr *links, int32_t r0, int32_t r1, int fx_shift, double inflation, double pruning,
                     hhx_csr **out, int64_t *n_products, int64_t *nnz_expanded);

/* ---------------------------------------------------------------- a12: interpret_result :2065-2095
 * Array half: attractors (ascending) = rows with a non-zero diagonal; members of attractor a =
 * rows of CSR(T) holding column a (ascending).  att[n], att_ptr[n+1], members[nnz] are host
 * buffers.  The set-of-tuples and the partition check (:2084-2095) stay in Python. */
int hhx_interpret(const hhx_csr *m, int32_t *att, int32_t *att_ptr, int32_t *members, int32_t *n_att);

/* ---------------------------------------------------------------- S4: dict_to_matrix :310-373
 * Input: the flank table in dict-insertion order (frag id pairs + value, device or host per
 * `on_device`), frag_set membership flags, n_rest = number of link-less members of frag_set (their
 * indices follow the linked ones; their order is CPython's set order and stays in Python).
 * Output: CSR(T) with self-loops (value 1) when add_self_loops, and frag_index[n_frag] (host; -1 =
 * not linked).  Index assignment = first appearance scanning the items in order, i before j. */
int hhx_dict_to_matrix(int64_t n_keys, const int32_t *frag_i, const int32_t *frag_j, const double *value,
                       int on_device, int32_t n_frag, const uint8_t *in_set_host, int32_t n_rest,
                       int add_self_loops, int32_t *frag_index_host, int32_t *n_linked, hhx_csr **out);

/* ---------------------------------------------------------------- a6: link weights on the flank table
 * The in-place dict rewrites between the ingest and dict_to_matrix, on the (frag_i, frag_j, value) arrays in dict order
 * (host arrays, or the device arrays of hhx_ingest_flank_device with on_device = 1: the weights then never leave HBM
 * on their way into hhx_dict_to_matrix).  value is float64 as in the reference (cast to float32 at :368).
 *   mode 0  normalize_by_nlinks :718-724: value /= (links[i] * links[j]) ** 0.5; per_frag_host = frag_link_dict totals [n_frag]
 *   mode 1  normalize_by_length :727-738 (dead code in the reference): value /= (fl_i / 1e6) * (fl_j / 1e6),
 *           fl = min(length, param), param = 2 * flank in bp; per_frag_host = fragment lengths [n_frag]
 *   mode 2  reduce_inter_hap_HiC_links :695-707: value -= value * param (param = phasing_weight) where tag_host[i] != tag_host[j]
 *           (the haplotype tag read_depth_dict[frag][0] mapped to integers); *n_zero = number of entries that reached 0 —
 *           the reference deletes those from the dict, the caller compacts them out. */
int hhx_link_weights(int64_t n_keys, const int32_t *frag_i, const int32_t *frag_j, double *value, int on_device, int mode,
                     int32_t n_frag, const int64_t *per_frag_host, const int32_t *tag_host, double param, int64_t *n_zero);

/* ---------------------------------------------------------------- f3: reassign's per-group link sums
 * HapHiC_reassign.py parse_link_dict :217-263 (normalize_by_nlinks = False): for every contig pair (i, j) of full_link_dict,
 * in dict order, sums[i][group[j]] += links and sums[j][group[i]] += links (group -1 = 'ungrouped': skipped).  links are
 * integer counts (64-bit integer adds: exact, order-free).  sums_host / first_host: [n_ctg][n_groups]; first = dict
 * position (2 * key index + side) of the first contribution to the cell, -1 (all bits set) if none — the order in which
 * the reference's inner dicts received their keys. */
int hhx_group_link_sums(int64_t n_keys, const int32_t *frag_i, const int32_t *frag_j, const int64_t *links, int32_t n_ctg,
                        const int32_t *group_host, int32_t n_groups, int64_t *sums_host, int64_t *first_host);

/* ---------------------------------------------------------------- a5: restriction-site counts
 * count_RE_sites :75-84 for many segments of one sequence buffer (host bytes, letter case as the caller's
 * parse_fasta :87-113 leaves it): counts[s] = sum over sites of seq[off[s] : off[s]+len[s]].count(site),
 * Python str.count semantics (non-overlapping, leftmost first).  `sites` holds the n_sites patterns after
 * parse_RE_sites' N expansion (:56-72), concatenated; site_len their lengths (<= 32).  Segments may overlap:
 * stat_fragments :188-296 asks for whole contigs, bins and the flank prefix / suffix of either. */
int hhx_count_re_sites(const uint8_t *seq_host, int64_t seq_len, int64_t n_seg, const int64_t *seg_off,
                       const int64_t *seg_len, int32_t n_sites, const uint8_t *sites, const int32_t *site_len,
                       int64_t *counts_host);

/* ---------------------------------------------------------------- f1: filter_fragments rank sums :866-892
 * m: the link matrix WITHOUT self loops (dict_to_matrix(flank_link_dict, filtered_frags) :868).  The reference
 * ranks the fragments of every dense row by links descending (stable: ties and link-less fragments by index),
 * takes the topN of fragment f and sums min(rank_a(b), rank_b(a)) over the unordered pairs of that top list.
 * rank_sum_host[n]: the statistic for every row, computed on the sparse rows (no dense matrix, no sort). */
int hhx_rank_sums(const hhx_csr *m, int topN, int64_t *rank_sum_host);

/* ---------------------------------------------------------------- S5: ingest
 * parse_alignments_for_ctgs :1596-1655 (bins = 0) and parse_alignments :1658-1752 (bins = 1) on
 * integer ids.  The Python shim maps names to ids once (haphic_amd/cluster.py: FragTable):
 *   contig id c in [0,n_ctg); ctg_rank[c] = rank of the contig NAME under Python string order
 *   (key orientation :1629); a split contig owns nbins consecutive fragment ids from ctg_frag0[c];
 *   frag_rank[f] = rank of the fragment NAME (:1720); frag_nx[f] = membership in Nx_frag_set.
 * A contig id < 0 (or >= n_ctg) = name absent from fa_dict (:1625).
 * Positions are 0-based int32 as yielded by the generators (:1556 :1593).
 */
typedef struct {
    int32_t n_ctg, n_frag;
    const int32_t *ctg_rank;     /* [n_ctg]  host */
    const int64_t *ctg_len;      /* [n_ctg]  host */
    const int32_t *ctg_frag0;    /* [n_ctg]  host */
    const uint8_t *ctg_split;    /* [n_ctg]  host */
    const int32_t *frag_rank;    /* [n_frag] host */
    const int64_t *frag_len;     /* [n_frag] host */
    const uint8_t *frag_nx;      /* [n_frag] host */
    int64_t bin_size;            /* only when bins */
    int64_t flank;               /* bp (args.flank * 1000) */
    int32_t bins;                /* 0: parse_alignments_for_ctgs, 1: parse_alignments */
    int32_t skip_intra;          /* 1: drop ref == mref like pairs_generator_inter_ctgs :1582 */
    int64_t expected_keys;       /* unused (kept for ABI stability): the tables are sized on the device */
} hhx_ingest_config;

int hhx_ingest_create(const hhx_ingest_config *cfg, hhx_ingest **out);
/* global stream ordinal of this handle's first pair: rank r of a multi-GPU job passes the number of pairs
 * held by ranks < r, so that first-seen ordinals are comparable across ranks.  Before the first push. */
int hhx_ingest_set_ordinal_base(hhx_ingest *h, int64_t base);
/* one batch of read pairs in stream order (at most 2^32 - 2 per call); id/pos arrays are device
 * (on_device=1) or host memory.  Each push is aggregated on the device into one run of distinct keys. */
int hhx_ingest_push(hhx_ingest *h, int64_t n_pairs, const int32_t *id1, const int32_t *pos1,
                    const int32_t *id2, const int32_t *pos2, int on_device);
/* the same with 64-bit positions: contigs of 2^31 bp and more, where the reference switches its coordinate arrays to int64
 * (determine_int_type :116-147).  With hhx_ingest_keep_pairs on, positions must stay below 2^32 - 1 (32-bit side records). */
int hhx_ingest_push64(hhx_ingest *h, int64_t n_pairs, const int32_t *id1, const int64_t *pos1, const int32_t *id2,
                      const int64_t *pos2, int on_device);
/* close the stream: merges the runs into one table per dict; the numbers of keys are returned */
int hhx_ingest_finalize(hhx_ingest *h, int64_t *n_full_keys, int64_t *n_flank_keys);
/* host copies, in dict insertion order: full_link_dict keys/counts, the HT_link_dict counts of each
 * contig pair ([HH, HT, TH, TT] :404-416), flank_link_dict keys/counts, per-fragment flank link totals
 * (frag_link_dict).  Any pointer may be NULL.  The insertion order is materialised on the first call. */
int hhx_ingest_fetch(hhx_ingest *h, int32_t *full_i, int32_t *full_j, int64_t *full_cnt, int64_t *ht_cnt,
                     int32_t *flank_i, int32_t *flank_j, int64_t *flank_cnt, int64_t *frag_links);
/* device-resident flank table (insertion order) for hhx_dict_to_matrix(on_device=1) */
int hhx_ingest_flank_device(hhx_ingest *h, void **dev_frag_i, void **dev_frag_j, void **dev_value_f64);
int hhx_ingest_flank_count_device(hhx_ingest *h, void **dev_count_i64);   /* same rows, int64 counts */
/* dict_to_matrix :310-373 fused onto the handle's flank table (hhx_dict_to_matrix semantics; the dict
 * insertion order is taken from the first-seen ordinals and never materialised).  n_rest < 0: every
 * link-less member of in_set gets a trailing index (their relative order is CPython's set order and is
 * decided by the Python caller). */
int hhx_ingest_link_matrix(hhx_ingest *h, const uint8_t *in_set_host, int32_t n_rest, int add_self_loops,
                           int32_t *frag_index_host, int32_t *n_linked, hhx_csr **out);
/* Side products that need the read pairs themselves.  hhx_ingest_keep_pairs(h, 1) before the first
 * push keeps, for every pair counted in full_link_dict, its oriented 1-based contig coordinates.  After finalize,
 * hhx_ingest_fetch_pairs returns, per contig pair in dict insertion order (the order of hhx_ingest_fetch):
 *   clm[4 * n]: update_clm_dict :395-401, four distances per read pair, read pairs in stream order;
 *               clm_ptr[k] .. clm_ptr[k+1] are the READ PAIRS of key k (multiply by 4 for the array offset);
 *   crd[2 * m]: record_coord_pairs :454-459, the first max_read_pairs (coord_i, coord_j) of every key.
 * Host buffers: clm_ptr, crd_ptr [n_full + 1]; clm [4 * sum(full_cnt)]; crd [2 * sum(min(full_cnt, max))]. */
int hhx_ingest_keep_pairs(hhx_ingest *h, int on);
int hhx_ingest_fetch_pairs(hhx_ingest *h, int64_t max_read_pairs, int64_t *clm_ptr, int64_t *clm, int64_t *crd_ptr,
                           int64_t *crd);
/* ------------------------------------------------------------------ multi-GPU build of the link matrix
 * dict_to_matrix :310-373 over a pair stream that is split into per-rank chunks.  The matrix index of a fragment is
 * the rank of its first position (2 * first ordinal + side) over the WHOLE stream — a minimum, so the ranks
 * all-reduce(min) hhx_shard_first's array, rank it with hhx_rank_first (same result everywhere), and exchange matrix
 * entries by row owner (all-to-all(v) of hhx_shard_emit's arrays, counts[] entries for each owner) instead of whole
 * tables; hhx_rows_from_entries adds up the counts of equal (row, column) received from different chunks and writes
 * the owner's CSR row block [r0, r1) x shape (self loops :362-364, link-less rows :357-359 at the tail).
 * Entries: w0 = row << 29 | column, w1 = count (uint64 each).  first: int64[n_frag], INT64_MAX = no entry here. */
typedef struct hhx_shard hhx_shard;
int hhx_shard_create(hhx_ingest *h, const uint8_t *in_set, hhx_shard **out);           /* after hhx_ingest_finalize */
int hhx_shard_first(hhx_shard *s, void **first_dev);
int hhx_rank_first(int32_t n_frag, const void *first_dev, void *frag_index_dev, int32_t *n_linked);
int hhx_shard_emit(hhx_shard *s, const void *frag_index_dev, int32_t n_bounds, const int32_t *row_bounds,
                   void **w0_dev, void **w1_dev, int64_t *counts);
int hhx_rows_from_entries(int64_t n, const void *w0_dev, const void *w1_dev, int32_t r0, int32_t r1, int32_t shape,
                          int add_self_loops, hhx_csr **out);
/* the same row block from the received entries AS THE ALL-TO-ALL(V) DELIVERS THEM: n_runs runs (one per source rank; run k =
 * entries [run_off[k], run_off[k + 1]), run_off a host array), each in row order as hhx_shard_emit wrote it — no partition pass */
int hhx_rows_from_runs(int32_t n_runs, const int64_t *run_off, const void *w0_dev, const void *w1_dev, int32_t r0, int32_t r1, int32_t shape,
                       int add_self_loops, hhx_csr **out);
int hhx_shard_destroy(hhx_shard *s);

/* ------------------------------------------------------------------ a1: .pairs text -> id / position arrays
 * pairs_generator :1539-1559 and pairs_generator_inter_ctgs :1562-1583: skip blank and '#' lines, split on
 * whitespace, (ref, pos, mref, mpos) = (cols[1], int(cols[2]) - 1, cols[3], int(cols[4]) - 1), and the two
 * alignments.bed records per line (:1557).  Names are resolved against the FASTA names given at creation
 * (n_names strings, concatenated, name_off[n_names + 1]); a name that is not among them, and every skipped
 * line, yields id -1 — which hhx_ingest_push drops (:1610 / :1702) — so line k of the chunk is element k of
 * the arrays.  A chunk must hold whole lines.  A line with fewer than 5 columns or a malformed integer fails
 * the call with "IndexError: ..." / "ValueError: ..." in hhx_last_error(), as the reference raises.  The
 * arrays (device pointers, int32) stay valid until the next parse on the same parser. */
typedef struct hhx_pairs_parser hhx_pairs_parser;
int hhx_pairs_parser_create(int32_t n_names, const uint8_t *names, const int64_t *name_off, hhx_pairs_parser **out);
int hhx_pairs_parse(hhx_pairs_parser *p, const uint8_t *text, int64_t n_bytes, int on_device, int want_bed,
                    int64_t *n_lines, int64_t *bed_bytes);
int hhx_pairs_parser_arrays(hhx_pairs_parser *p, void **id1, void **pos1, void **id2, void **pos2, void **bed);
int hhx_pairs_parser_fetch(hhx_pairs_parser *p, int32_t *id1, int32_t *pos1, int32_t *id2, int32_t *pos2, uint8_t *bed);
/* wide mode: positions as int64 from the next parse on (contigs beyond 2^31 bp, :116-147): hhx_pairs_parser_arrays then hands out
 * int64 position arrays (for hhx_ingest_push64), hhx_pairs_parser_fetch64 copies them; ids, BED bytes and errors as before */
int hhx_pairs_parser_set_wide(hhx_pairs_parser *p, int on);
int hhx_pairs_parser_fetch64(hhx_pairs_parser *p, int32_t *id1, int64_t *pos1, int32_t *id2, int64_t *pos2, uint8_t *bed);
/* the alignments.bed bytes of the last parse in pinned host memory owned by the parser: two buffers used in turn, so the
 * pointer stays valid until the SECOND following call (the caller writes buffer k to the file while chunk k + 1 is parsed) */
int hhx_pairs_parser_bed_host(hhx_pairs_parser *p, void **host, int64_t *n_bytes);
int hhx_pairs_parser_destroy(hhx_pairs_parser *p);
/* the .pairs file as chunks of whole lines in PINNED host memory (what pairs_generator* :1543 / :1566 iterate line by line): read with pread() by
 * n_threads (<= 0: 4) threads into one of two buffers while the caller tokenises the other; a chunk holds at most ~2 x chunk_bytes and ends after its
 * last line break.  hhx_text_reader_next: *host is valid until the next call; *n_bytes == 0 at the end of the file.  hhx_pairs_parse(on_device = 0)
 * takes such a chunk to the device at PCIe rate. */
typedef struct hhx_text_reader hhx_text_reader;
int hhx_text_reader_open(const char *path, int64_t chunk_bytes, int n_threads, hhx_text_reader **out);
/* the same over a bgzipped file (`bgzipped_pairs`: gzip.open in the reference :1541 / :1564): BGZF blocks inflated by n_threads threads straight into the
 * pinned buffer.  A file that is not BGZF (a plain gzip stream has no block boundaries to inflate in parallel) is refused: the caller keeps its gzip reader. */
int hhx_text_reader_open_bgzf(const char *path, int64_t chunk_bytes, int n_threads, hhx_text_reader **out);
int hhx_text_reader_next(hhx_text_reader *r, const uint8_t **host, int64_t *n_bytes);
int hhx_text_reader_close(hhx_text_reader *r);
/* measurement only — the writer counterpart of hhx_pairs_parse for synthetic read pairs ("pairs written as .pairs text"): line k =
 * "r{first_read + k}\t{names[id1[k]]}\t{pos1[k] + 1}\t{names[id2[k]]}\t{pos2[k] + 1}\t+\t-\n" from device arrays (ids must be valid: 0 <= id < n_names).
 * *n_bytes = the size of the text; dev_text == NULL: only that.  Not on the product path. */
int hhx_pairs_format(hhx_pairs_parser *p, int64_t n, const int32_t *dev_id1, const int32_t *dev_pos1, const int32_t *dev_id2, const int32_t *dev_pos2,
                     int64_t first_read, uint8_t *dev_text, int64_t capacity, int64_t *n_bytes);

/* ------------------------------------------------------------------ f4: BAM front end
 * bam_generator :1586-1593 = pysam.AlignmentFile(bam, threads=..., format_options=[b'filter=...']) yielding
 * (reference_name, next_reference_name, reference_start, next_reference_start) per record that passes the htslib filter
 * ('flag.read1' :2837 :2855, 'flag.read1 && refid != mrefid' :2862).  hhx_bam_open reads the BGZF container and the BAM
 * header (SAM text for check_sorting_order :1347-1359; reference names, concatenated, name_off[n_ref + 1]).
 * hhx_bam_next inflates the next batch of BGZF blocks (host threads), walks the record lengths and decodes the batch on the
 * device: record k of the batch = element k of four int32 device arrays (valid until the next call), ids mapped through
 * ref_to_ctg_host[n_ref] (BAM reference id -> contig id of the FASTA, -1 = not in fa_dict), records that fail the filter
 * (flag & need_flags != need_flags; or refID == next_refID when drop_same_ref) carry id -2 in both columns, unmapped ends
 * and references outside the map id -1 — hhx_ingest_push drops every negative id, as `ref not in fa_dict` does
 * (:1610 / :1702).  *n_records == 0: end of file. */
typedef struct hhx_bam hhx_bam;
int hhx_bam_open(const char *path, int threads, hhx_bam **out);
int hhx_bam_header(hhx_bam *b, int32_t *n_ref, const char **text, int64_t *text_len, const char **names, const int64_t **name_off);
int hhx_bam_next(hhx_bam *b, int need_flags, int drop_same_ref, int32_t n_ref, const int32_t *ref_to_ctg_host, int64_t max_inflated_bytes,
                 int64_t *n_records, void **id1_dev, void **pos1_dev, void **id2_dev, void **pos2_dev);
int hhx_bam_fetch(hhx_bam *b, int32_t *id1, int32_t *pos1, int32_t *id2, int32_t *pos2);   /* host copies of the last batch */
int hhx_bam_close(hhx_bam *b);

/* ------------------------------------------------------------------ f4: `haphic plot` contact map
 * HapHiC_plot.py parse_pairs :153-202 / parse_bam :205-245: contact_matrix[bin(ref, pos), bin(mref, mpos)] += 1 per read
 * pair whose two contigs are in ctg_set, with convert_group_bin_id :155-168 as the position -> total scaffold bin lookup.
 * The reference's dicts, flattened (haphic_amd/plot.py does this from ctg_dict / ctg_aln_dict / group_to_total_bin_dict):
 *   in_set[c]                       contig c is in ctg_set (:145-148)
 *   aln_ptr[c] .. aln_ptr[c + 1]    slots of the contig's alignment bins 0, 1, ... (slot = aln_ptr[c] + (pos - 1) // bin_size)
 *   list_ptr[slot] .. [slot + 1]    the closed ranges ctg_aln_dict[ctg][bin] lists, in list order; an empty list = no such key
 *   seg_lo/seg_hi/seg_bin[k]        range on the raw contig (1-based, closed) and group_to_total_bin_dict[ctg_dict[ctg][range]],
 *                                   -1 when the scaffold is not in group_list (the pair is skipped, :161-162)
 * hhx_contact_map_push adds a batch (ids < 0 are skipped: unknown names, filtered BAM records, unmapped mates;
 * pos_offset is added to every position: 0 for .pairs text, 1 for BAM's 0-based reference_start :232 :236).
 * *bad = -1, or 2 * k + side of the first pair k whose position has no alignment bin — where the reference raises
 * 'Cannot find alignment position' (:165-168); the counts of that batch are then meaningless, as in the reference.
 * hhx_contact_map_fetch: the n_total_bins x n_total_bins int64 matrix, row major (numpy dtype=int, :144). */
typedef struct hhx_contact_map hhx_contact_map;
int hhx_contact_map_create(int32_t n_ctg, const uint8_t *in_set, const int64_t *aln_ptr, const int32_t *list_ptr, int64_t n_list,
                           const int32_t *seg_lo, const int32_t *seg_hi, const int32_t *seg_bin, int32_t bin_size, int32_t n_total_bins,
                           hhx_contact_map **out);
int hhx_contact_map_push(hhx_contact_map *m, int64_t n_pairs, const int32_t *id1, const int32_t *pos1, const int32_t *id2, const int32_t *pos2,
                         int on_device, int32_t pos_offset, int64_t *bad);
int hhx_contact_map_fetch(hhx_contact_map *m, int64_t *cells_host);
int hhx_contact_map_device(hhx_contact_map *m, void **cells_dev_i64, int32_t *n_total_bins);
int hhx_contact_map_destroy(hhx_contact_map *m);

/* HT_link_dict's insertion order (update_HT_link_dict :404-416): first[4 * k + q] = stream position (among the pairs that
 * entered full_link_dict) of the first read pair of contig pair k (dict order of hhx_ingest_fetch) in quadrant
 * q = [HH, HT, TH, TT], INT64_MAX if the quadrant is empty.  Needs hhx_ingest_keep_pairs. */
int hhx_ingest_fetch_ht_order(hhx_ingest *h, int64_t *first);
/* ctg_pair_to_frag :1731-1733 (split contigs + --remove_allelic_links): every distinct oriented fragment pair the
 * stream produced, whatever its flank / Nx status.  Request before the first push; fetch after finalize (call with
 * NULL arrays for the size). */
int hhx_ingest_keep_frag_pairs(hhx_ingest *h, int on);
int hhx_ingest_fetch_frag_pairs(hhx_ingest *h, int64_t *n_pairs, int32_t *frag_i, int32_t *frag_j);
int hhx_ingest_destroy(hhx_ingest *h);

/* Multi-GPU exchange step (ingest).  The aggregated table of a finalized handle, device
 * SoA in no particular order: key = (i << 29) | j, first-seen global ordinals of the key in
 * full_link_dict / flank_link_dict (UINT64_MAX = never), HT counts [n][4], flank count.
 * which = 0: contig-pair table, 1: fragment-pair flank table (the same table unless contigs are split).
 * hhx_ingest_push_table feeds such rows (e.g. the all-gathered tables of every rank) into another handle;
 * its finalize merges them: counts add, ordinals take the minimum — exactly the reference loop run over
 * the concatenated stream. */
int hhx_ingest_table_device(hhx_ingest *h, int which, int64_t *n_rows, void **dev_key_u64, void **dev_ord_full_u64,
                            void **dev_ord_flank_u64, void **dev_ht_u32x4, void **dev_flank_u32);
int hhx_ingest_push_table(hhx_ingest *h, int which, int64_t n_rows, const uint64_t *dev_key, const uint64_t *dev_ord_full,
                          const uint64_t *dev_ord_flank, const uint32_t *dev_ht, const uint32_t *dev_flank);


/* ---------------------------------------------------------------- the files run() writes from the S5 containers (:2879-2929)
 * With the link tables held as arrays (haphic_amd/containers.py) the reference's writers have array forms:
 *
 * output_clm :376-392 — paired_links.clm from the read pairs kept in HBM (hhx_ingest_keep_pairs): for every contig pair with at
 * least two read pairs (:385), in full_link_dict order, four lines "{ctg_i}{+|-} {ctg_j}{+|-}\t{2 * links}\t{d d d d ...}\n",
 * the distances of update_clm_dict :395-401 for that orientation ascending, each written twice (:388).  Grouping, sorting and
 * the text are made on the device; contig names = UTF-8 bytes back to back (host), name k = [name_off[k], name_off[k + 1]).
 * Returns the numbers of lines and bytes written.  Fails (and leaves a partial file) if a read position lies beyond its contig. */
int hhx_ingest_write_clm(hhx_ingest *h, const char *path, const uint8_t *names_blob, const int64_t *name_off, int64_t *n_lines,
                         int64_t *n_bytes);
/* HT_link_dict (update_HT_link_dict :404-416) as items in dict insertion order, sorted on the device: item t = the t-th (contig
 * pair, quadrant) entry to enter the dict, i.e. ordered by the stream position of its first read pair (hhx_ingest_fetch_ht_order);
 * name_i / name_j = 2 * contig id + (1 if that end is '_T', 0 for '_H'), count = its links.  *n_items is always set; the arrays
 * (host, [*n_items] from a first call with null arrays) are optional.  Needs hhx_ingest_keep_pairs. */
int hhx_ingest_fetch_ht_items(hhx_ingest *h, int64_t *n_items, int32_t *name_i, int32_t *name_j, int64_t *count);
/* the float64 values of the flank table in dict order (host buffer [n_flank_keys]): the counts, or what hhx_link_weights
 * (on_device, over hhx_ingest_flank_device's arrays) left there — normalize_by_nlinks :718-724 on the resident table */
int hhx_ingest_fetch_flank_values(hhx_ingest *h, double *value);
/* output_pickle :710-715 for a link dict given as arrays — full_links.pkl, HT_links.pkl: a protocol-4 pickle of
 * `collections.defaultdict(int)` {(names[name_i[k]], names[name_j[k]]): count[k]}, keys in array order (= dict insertion order),
 * written by host code without a Python object per key; names as for hhx_ingest_write_clm.  pickle.load gives the dict the
 * reference's loops :1605-1615 build.  All pointers host. */
int hhx_write_link_pickle(const char *path, int64_t n_keys, const int32_t *name_i, const int32_t *name_j, const int64_t *count,
                          int32_t n_names, const uint8_t *names_blob, const int64_t *name_off, int64_t *n_bytes);


/* ---------------------------------------------------------------- the same files, off the caller's critical path
 * run() :2879 / :2888 / :2929 writes HT_links.pkl, paired_links.clm and full_links.pkl between its seams and reads none of them again
 * (the next readers are `haphic sort` / `haphic reassign`, other processes).  The *_async entry points check their arguments, refuse
 * what the writer would refuse (a read position beyond its contig's end) and open the file on the CALLER's thread, then queue the work
 * on a host thread owned by the library — two lanes (the byte sinks below; everything else), each running its jobs in submission order on a
 * non-blocking stream of its own with a memory-pool arena of its own, so the device half of a job (grouping / sorting / formatting, hhx_ingest_write_clm; ordering the HT items,
 * hhx_ingest_fetch_ht_items) overlaps the caller's next kernels — and return at once.  hhx_files_join waits for every queued file and
 * returns non-zero with the first failure as hhx_last_error() (*n_failed = how many files failed; the failures are forgotten after the
 * call); hhx_ingest_destroy waits for the jobs that read its handle.  A process that exits without hhx_files_join still gets its
 * files: the library's destructor works the queue off.
 *
 * hhx_ingest_write_clm_async: output_clm :376-392.  drop_pairs_after != 0: the kept read pairs (16 B per pair of HBM) are released when
 * the file is complete — afterwards hhx_ingest_fetch_pairs / _ht_order / _ht_items / _write_clm on this handle fail (with that message).
 * hhx_ingest_write_link_pickle_async: output_pickle :710-715 of a table of the handle — which = 0 full_link_dict (names = contigs),
 * 1 HT_link_dict (names = [c0 + "_H", c0 + "_T", c1 + "_H", ...], items ordered on the device), 2 flank_link_dict with its integer counts
 * (names = fragments); the writer thread fetches the items into host memory of its own.
 * hhx_write_link_pickle_async: hhx_write_link_pickle of arrays the CALLER owns: name_i / name_j / count must stay valid and unchanged until
 * hhx_files_join returns (names_blob / name_off are copied). */
int hhx_ingest_write_clm_async(hhx_ingest *h, const char *path, const uint8_t *names_blob, const int64_t *name_off, int drop_pairs_after);
int hhx_ingest_write_link_pickle_async(hhx_ingest *h, int which, const char *path, int32_t n_names, const uint8_t *names_blob,
                                       const int64_t *name_off);
int hhx_write_link_pickle_async(const char *path, int64_t n_keys, const int32_t *name_i, const int32_t *name_j, const int64_t *count,
                                int32_t n_names, const uint8_t *names_blob, const int64_t *name_off);
/* alignments.bed (pairs_generator* :1549-1557: two records per read pair, written inside the generator's loop, read by nothing in run()), deferred:
 * a byte sink is a file fed from DEVICE memory through the same writer thread.  The producer reserves room in the sink's ring of HBM slabs
 * (hhx_byte_sink_reserve: returns at once while fewer than hbm_budget_bytes are waiting to be written — <= 0: a quarter of the device, or
 * HHX_BED_HBM_GB — and holds the caller at the writer's pace beyond that), launches the kernel that fills it on its own stream, and commits
 * (hhx_byte_sink_commit: the range is queued; the writer waits for the kernel through an event).  expected_bytes (0: unknown) sizes the slabs.
 * hhx_pairs_parser_set_bed_sink makes hhx_pairs_parse(want_bed) do exactly that with the BED records of every chunk.  hhx_byte_sink_close
 * queues the close and frees the handle when it has run: the file is complete after hhx_files_join. */
typedef struct hhx_byte_sink hhx_byte_sink;
int hhx_byte_sink_open(const char *path, int64_t hbm_budget_bytes, int64_t expected_bytes, hhx_byte_sink **out);
int hhx_byte_sink_reserve(hhx_byte_sink *sink, int64_t n_bytes, void **dev);
int hhx_byte_sink_commit(hhx_byte_sink *sink, void *dev, int64_t n_bytes);
int hhx_byte_sink_close(hhx_byte_sink *sink, int64_t *n_bytes_pushed);
int hhx_pairs_parser_set_bed_sink(hhx_pairs_parser *p, hhx_byte_sink *sink);
int hhx_files_pending(int64_t *n_pending, int64_t *n_done);
int hhx_files_join(int64_t *n_failed);

#ifdef __cplusplus
}
#endif
#endif /* HAPHIC_HIP_H */
