/*
 * mobgt_hip.h -- C ABI of libmobgt_hip.so, the MI355X (gfx950) hot path of MobGT.
 *
 * This is the drop-in boundary.  The reference (Yukayo/MobGT) is pure
 * Python/PyTorch + one Cython file, so "the reference's FFI for this path" is the set of Python
 * call sites listed next to each entry point below; the binding a maintainer adds is a ctypes
 * stub (INTEGRATION.md shows it; mobgt_amd/_lib.py is the one this repo ships).
 *
 * Conventions
 *   - every entry point returns 0 on success, a hipError_t (>0) from the launch, or a negative
 *     MOBGT_E* code for an argument the kernels cannot take;
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every buffer;
 *   - no allocation, no synchronisation, no host<->device copy inside: each call enqueues kernels
 *     on `stream` (a hipStream_t passed as void*) and returns, so calls can be captured in a hipGraph;
 *   - `dtype` arguments: MOBGT_F32 = 0, MOBGT_BF16 = 1; index tensors: MOBGT_I64 / I32 / I16 / U8;
 *   - G graphs, H heads, T = N+1 tokens (token 0 = graph token), N padded nodes, d = head width
 *     (16, 24 or 32), D = multi_hop_max_dist (<= 32), F edge-feature columns.
 */
#ifndef MOBGT_HIP_H
#define MOBGT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOBGT_F32 0
#define MOBGT_BF16 1

#define MOBGT_I64 0
#define MOBGT_I32 1
#define MOBGT_I16 2
#define MOBGT_U8 3

#define MOBGT_EBADDIM (-1)   /* unsupported head width / hop count / size            */
#define MOBGT_EALIGN (-2)    /* pointer or stride violates the documented alignment  */
#define MOBGT_EDTYPE (-3)    /* unknown dtype code                                   */

/* Library / device identification (no GPU work). */
/* Bumped on every incompatible change of a signature below (round 5 inserted `out_lo` into the attention entry points under
 * version 1 -- ADVICE r5; version 2 = that ABI; version 3 = round 6).  mobgt_amd/_lib.py refuses a library whose
 * version differs from the one its SIGNATURES table was written for. */
#define MOBGT_ABI_VERSION 3
int mobgt_abi_version(void);
const char* mobgt_build_info(void);

/* ------------------------------------------------------------------------------------------------
 * Bias-fused multi-head attention.  Replaces the body of MultiHeadAttention.forward between the
 * three input projections and the output projection:
 *     graphormer/model.py:436-455   (view/transpose, q*scale, matmul, +attn_bias, softmax, dropout,
 *     graphormer/model_fqandtoyo.py:1687-1706    matmul, transpose/contiguous/view)
 *
 * q,k,v,out : [G, T, H*d] row-major with row strides ldq/ldk/ldv/ldo (elements); head h occupies
 *             columns [h*d, (h+1)*d).  dtype `io_dtype` (f32 or bf16); 16-byte aligned rows.
 * bias      : [G, H, T, ld_bias] `bias_dtype`, bias[g,h,i,j] added unscaled to the score of query i,
 *             key j; ld_bias % 64 == 0, ld_bias >= roundup(T,64) (rows are fetched as whole 64-key segments);
 *             columns >= T are ignored by the forward / dQ pass and must hold -inf in bias_t (the pack kernels write them).
 *             -inf entries are honoured (probability exactly 0).
 * out_lo    : bf16 I/O only (ignored for f32; may be NULL): [G, T, H*d] bf16, row stride ldo -- the rounding residual
 *             O - bf16(O) of `out`.  The backward's delta = rowsum(dO * O) needs O beyond bf16: its error is 2^-9 of |dO||O|,
 *             un-cancelled, against a dS whose rows sum to zero (csrc/attn.hip header, "consistent softmax").
 * lse       : [G, H, T] f32 out, natural-log sum of exp of the biased scores (needed by bwd).
 * scale     : q is multiplied by it BEFORE the dot product (reference: att_size ** -0.5).
 * dropout_p : attention dropout on the probabilities (model.py:451); 0 disables.  The keep mask is
 *             a pure function of (seed, g, h, i, j) -- see mobgt_dropout_keep_host() -- so the backward
 *             regenerates it.  `seed_dev` (may be NULL) is a device uint64 added to `seed` (lets a
 *             captured graph advance the seed without re-capture).
 */
int mobgt_attn_bias_fwd(const void* q, const void* k, const void* v, const void* bias,
                        void* out, void* out_lo, float* lse,
                        int G, int H, int T, int d,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t ld_bias,
                        float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                        int io_dtype, int bias_dtype, void* stream);

/* Backward of the above (autograd of model.py:442-453).
 * bias_t : [G, H, T, ld_bias] the same bias with query/key transposed (bias_t[g,h,j,i] = bias[g,h,i,j]),
 *          produced by mobgt_build_bias / mobgt_bias_pack; read by the dK/dV pass.
 * out, out_lo : what the forward wrote (out_lo: bf16 I/O only, may be NULL -- the gradients of q / k then carry the
 *          un-cancelled rounding of `out`).
 * dout   : [G, T, H*d] (ldo), gradient of `out`.
 * dq,dk,dv : [G, T, H*d] `io_dtype`, row strides lddq/lddk/lddv; fully overwritten.
 * dbias  : [G, H, T, ld_bias] or NULL; columns >= T are left untouched.  The bias is shared by all L layers
 *          (model.py:207-208), so its gradient is a sum over layers; two ways to form it:
 *          dbias_dtype MOBGT_F32 : one f32 buffer; accumulate_dbias != 0: dbias += dS (read-modify-write),
 *                                  == 0: dbias = dS;
 *          dbias_dtype MOBGT_BF16: this layer's own bf16 slice, written once (dbias = dS; accumulate_dbias
 *                                  ignored); mobgt_build_bias_bwd sums the L slices.  A quarter of the f32
 *                                  path's HBM traffic per layer; dS is rounded to bf16 exactly as it is for
 *                                  the dQ / dK contractions.
 * delta  : [G, H, T] f32 workspace (kept in the signature; since round 5 every pass forms its own rowsum(dout*out) from the
 *          bf16 dout values its dP product multiplies).
 */
int mobgt_attn_bias_bwd(const void* q, const void* k, const void* v, const void* bias, const void* bias_t,
                        const void* out, const void* out_lo, const float* lse, const void* dout,
                        void* dq, void* dk, void* dv, void* dbias, float* delta,
                        int G, int H, int T, int d,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                        int64_t lddq, int64_t lddk, int64_t lddv, int64_t ld_bias,
                        float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                        int accumulate_dbias, int dbias_dtype, int io_dtype, int bias_dtype, void* stream);
/* The same backward with an f32 accumulator for dQ: dq_acc [G, T, H*d], 16-byte aligned, ALL ZEROS on entry and all zeros again
 * when the call's last launch has run (keep one per stream; never touch it in between).  With it, T > 64, bf16 I/O, a bf16 bias and
 * a bf16 dBias slice the gradients are formed in ONE pass over the bias (csrc/attn.hip: attn_bwd_one_kernel -- keys on the lanes,
 * 256 keys per 8-wave workgroup: S / P / dS once per pair, bias_t read once, dBias written once, rowsum(dO * O) formed inside the
 * pass; dQ is summed over the key blocks by f32 atomics, so it is then NOT bitwise reproducible from run to run, unlike the two
 * passes).  Any other configuration, dq_acc = null or MOBGT_ATTN_TWO_PASS=1 in the environment: exactly mobgt_attn_bias_bwd.
 * Replaces the autograd of graphormer/model.py:436-455 like that function. */
int mobgt_attn_bias_bwd_fused_z(const void* q, const void* k, const void* v, const void* bias, const void* bias_t,
                                const void* out, const void* out_lo, const float* lse, const void* dout, void* dq, void* dk, void* dv,
                                void* dbias, float* delta, int G, int H, int T, int d, int64_t ldq, int64_t ldk,
                                int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t ld_bias,
                                float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                int accumulate_dbias, int dbias_dtype, int io_dtype, int bias_dtype, float* dq_acc, void* stream);

/* Host-side statement of the dropout keep rule used by both kernels (for tests / replay).
 * Returns 1 if probability element (g,h,i,j) is kept. */
int mobgt_dropout_keep_host(uint64_t seed, int H, int T, int g, int h, int i, int j, float dropout_p);
/* The same for the whole [G,H,T,T] mask at once (host memory, 1 byte per element). */
int mobgt_attn_dropout_mask_host(uint64_t seed, int G, int H, int T, float dropout_p, uint8_t* out);
/* Host replay of the keep rule of every other dropout site of the step (nn.Dropout at model.py:476-488,
 * model_fqandtoyo.py:358, 1347, 1364, modelGNN.py:72): out[r*C + c] = 1 if element c of row (row0 + r) is kept;
 * `seed` = host seed + device step counter, `salt` = the site's constant. */
int mobgt_dropout_mask_host(uint64_t seed, uint32_t salt, int64_t row0, int64_t R, int C, float dropout_p, uint8_t* out);

/* ------------------------------------------------------------------------------------------------
 * Re-layout of a caller-supplied attention bias into the padded row-major + transposed pair the
 * attention kernels read.  Used when EncoderLayer/MultiHeadAttention are called with an arbitrary
 * `attn_bias` tensor (model.py:479-481), i.e. not produced by mobgt_build_bias.
 * src: [G,H,T,T] with element strides (s_g, s_h, s_i, s_j) (s_h may be 0 for a broadcast head dim).
 */
int mobgt_bias_pack(const void* src, int src_dtype, int64_t s_g, int64_t s_h, int64_t s_i, int64_t s_j,
                    void* bias, void* bias_t, int bias_dtype,
                    int G, int H, int T, int64_t ld_bias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention-bias assembly fused with the multi-hop edge-feature reduce.  Replaces
 *     graphormer/model.py:126-190            (stock: rel_pos + virtual-token column + edge term + 2*attn_bias)
 *     graphormer/model_fqandtoyo.py:1143-1216 (fq: additionally poi_pos; fp16 rounding points)
 *
 * attn_bias : [G, T, T] f32 (0 / -inf from the collator, collator.py:57-64); counted twice (model.py:190)
 * rel_pos   : [G, N, N] indices into rel_table [n_rel, H]    (dtype idx_dtype)
 * poi_pos   : [G, N, N] indices into poi_table [n_poi, H], or NULL (stock variant)
 * edge_input: [G, N, N, D_in, F] indices (edge_dtype) into the fused hop table; only the first
 *             D = min(D_in, multi_hop_max_dist) hops are read (model.py:163)
 * hop_table : [D, n_edge, H] f32 = sum_h' edge_encoder[e,h'] * edge_dis_encoder[d,h',h] (model.py:166-176),
 *             built by the host module with the fq variant's fp16 rounding points applied when asked;
 *             the per-pair reduce is  sum_d mean_f hop_table[d, edge_input[..,d,f], :] / spd  (model.py:180-182)
 * vdist     : [H] graph_token_virtual_distance.weight (added to column 0 of rows 1..N only, model.py:139-151)
 * bias, bias_t : outputs, [G, H, T, ld_bias] bias_dtype (see mobgt_attn_bias_fwd); columns T..ld_bias-1 = -inf
 */
int mobgt_build_bias(const float* attn_bias, const void* rel_pos, const void* poi_pos, const void* edge_input,
                     const float* rel_table, const float* poi_table, const float* hop_table, const float* vdist,
                     void* bias, void* bias_t,
                     int G, int N, int H, int D_in, int D, int F, int n_rel, int n_poi, int n_edge,
                     int64_t ld_bias, int idx_dtype, int edge_dtype, int bias_dtype, void* stream);

/* Backward of mobgt_build_bias: scatters the bias gradient (sum over layers) into the table gradients
 * (all f32, ACCUMULATED into -- zero them first):
 *   d_rel_table [n_rel,H], d_poi_table [n_poi,H] (or NULL), d_hop_table [D,n_edge,H], d_vdist [H].
 * dbias: dbias_dtype MOBGT_F32: one [G,H,T,ld_bias] f32 buffer already summed over layers (n_slices ignored);
 *        MOBGT_BF16: n_slices bf16 buffers of that shape, slice_stride ELEMENTS apart (one per layer, as written
 *        by mobgt_attn_bias_bwd), summed here in f32.
 * Entries where attn_bias is -inf carry no gradient and are skipped.
 */
int mobgt_build_bias_bwd(const void* dbias, int dbias_dtype, int n_slices, int64_t slice_stride,
                         const float* attn_bias, const void* rel_pos, const void* poi_pos,
                         const void* edge_input,
                         float* d_rel_table, float* d_poi_table, float* d_hop_table, float* d_vdist,
                         int G, int N, int H, int D_in, int D, int F, int n_rel, int n_poi, int n_edge,
                         int64_t ld_bias, int idx_dtype, int edge_dtype, void* stream);
/* Round 6: the persistent workgroups of the LONG-batch form of mobgt_build_bias_bwd (G (N+1)^2 >= 2^20 pairs: one 8-wave
 * workgroup per compute unit walks the pairs) for the launches that follow: 0 = one per compute unit (default), n = at most n.  A
 * caller that issues the launch on a side stream under other work (the tail of the backward pass: autograd of
 * model_fqandtoyo.py:1143-1216 needs nothing but the dBias slices) leaves a quarter of the units to that work; no effect on
 * results.  Host-side state, not thread-safe. */
int mobgt_build_bias_bwd_set_workgroups(int n);

/* Hop table of the multi-hop edge term (model.py:166-176; fq: model_fqandtoyo.py:1178-1198):
 *   table[d, e, h] = sum_h' edge_encoder[e, h'] * edge_dis_encoder[d, h', h]     [D, n_edge, H] f32
 * from edge_encoder.weight [n_edge, H] and edge_dis_encoder.weight viewed as [>= D, H, H].  fp16_roundtrip != 0
 * applies the fq variant's rounding points (operands and product rounded to fp16, fp32 accumulate).
 * Backward: d_edge_encoder [n_edge, H] (row 0 = padding_idx: written as zero) and d_edge_dis_encoder [D, H, H],
 * both overwritten; with fp16_roundtrip the gradients are rounded to fp16 where autograd would pass them back
 * through the reference's `.half()` casts. */
int mobgt_hop_table_fwd(const float* edge_encoder, const float* edge_dis_encoder, float* table, int D, int n_edge,
                        int H, int fp16_roundtrip, void* stream);
int mobgt_hop_table_bwd(const float* d_table, const float* edge_encoder, const float* edge_dis_encoder,
                        float* d_edge_encoder, float* d_edge_dis_encoder, int D, int n_edge, int H,
                        int fp16_roundtrip, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batched shortest-path preprocessing on the device.  Replaces, for a whole padded batch,
 *     graphormer/algos.pyx:9-54   floyd_warshall   (bit-exact M and path, 510 sentinel, k-sequential)
 *     graphormer/algos.pyx:57-96  get_all_edges + gen_edge_input (first D hops only; node-0 quirk kept)
 *     graphormer/wrapper.py:55-61,97-98 and the +1 / pad-0 shifts of collator.py:76-93
 *
 * counts   : [G, N, N] int32 transition counts (0 = no edge), rows/cols >= n_nodes[g] ignored
 * n_nodes  : [G] int32 real node count per graph
 * spd      : [G, N, N] int16 out: raw M (0..510) for real pairs, -1 for padding   (wrapper.py:61)
 * path     : [G, N, N] int16 out: raw path matrix, -1 for padding
 * rel_pos  : [G, N, N] int16 out: M+1 for real pairs, 0 for padding                (collator.py:76-83)
 * edge_input: [G, N, N, D, 1] uint8 out: hop feature+1 (= count+3 on an edge, 1 on a non-edge hop of a
 *             truncated path, 0 = no hop / padding)                                 (collator.py:86-93)
 * in_degree/out_degree: [G, N] int16 out: row-sum+1 / col-sum+1 of the 0/1 adjacency, 0 for padding
 *             (wrapper.py:97-98 naming kept; collator.py:11-18)
 * work     : scratch, mobgt_spd_workspace_bytes(G, N) bytes (272 < N <= 1088: per-row publication flags and rows of
 *            the multi-workgroup Floyd-Warshall; zeroed by the call itself)
 * Counts above 252 saturate the uint8 hop feature (the reference's edge tables have 128 rows).
 */
int64_t mobgt_spd_workspace_bytes(int G, int N);
/* Bound (in 100 MHz wall-clock ticks; < 0 restores the default, 200 ms) on how long a workgroup of the multi-workgroup
 * Floyd-Warshall waits for a row another workgroup publishes before its graph is handed to the single-workgroup redo
 * pass.  0 forces the redo pass for every long graph (tests). */
int mobgt_spd_set_spin_limit(int64_t ticks_100mhz);
int mobgt_spd_batched(const int32_t* counts, const int32_t* n_nodes,
                      int16_t* spd, int16_t* path, int16_t* rel_pos, uint8_t* edge_input,
                      int16_t* in_degree, int16_t* out_degree, void* work,
                      int G, int N, int D, void* stream);

/* The elementwise remainder of collator_foursquare / collator_gowalla on the device (collator.py:57-64, 354-358, 428-437):
 * attn_bias [G, N+1, N+1] f32 = 0 for key columns 0..n_nodes[g], -inf beyond (and -inf where spd >= rel_pos_max when
 * rel_pos_max <= 510); poi_pos [G, N, N] int16 = bin_table[x_i, x_j] for real pairs (x != 0), 0 otherwise.
 * x [G, N] int32 POI ids (0 = pad); spd [G, N, N] int16 (mobgt_spd_batched); bin_table [*, ld_bin] int16 or NULL (then
 * poi_pos = 0). */
int mobgt_collate_finish(const int32_t* x, const int32_t* n_nodes, const int16_t* spd, const int16_t* bin_table,
                         int64_t ld_bin, int rel_pos_max, float* attn_bias, int16_t* poi_pos, int G, int N, void* stream);


/* Single-graph entry points with the reference's own call signatures, for the per-item drop-in path
 * (wrapper.py:55-60 calls algos.floyd_warshall(adj) and algos.gen_edge_input(max_dist, path, edge_feat)).
 *   mobgt_floyd_warshall  <- graphormer/algos.pyx:9-54   adj [n,n] int64 (non-zero = edge) -> M, path [n,n] int64
 *   mobgt_gen_edge_input  <- graphormer/algos.pyx:65-96  path [n,n] int64 (ANY path matrix), edge_feat [n,n,F] int64
 *                            -> out [n,n,max_dist,F] float32, -1 fill.  *err_flag (device int) is set non-zero
 *                            when a path has more hops than max_dist (the reference raises IndexError) or the
 *                            path matrix does not terminate.
 * `work` for mobgt_floyd_warshall: mobgt_floyd_warshall_workspace_bytes(n) bytes of device scratch.
 */
int64_t mobgt_floyd_warshall_workspace_bytes(int n);
int mobgt_floyd_warshall(const int64_t* adj, int n, int64_t* M, int64_t* path, void* work, void* stream);
/* graphormer/algos.pyx:57-62 for one pair: out_nodes [n+2] int32 receives the intermediate nodes of the
 * path i -> j, *out_len their count (-1: the path matrix does not terminate); work: 2n+4 int32. */
int mobgt_get_all_edges(const int64_t* path, int n, int i, int j, int32_t* out_nodes, int32_t* out_len,
                        int32_t* work, void* stream);
int mobgt_gen_edge_input(int max_dist, const int64_t* path, const int64_t* edge_feat, int n, int F,
                         float* out, int* err_flag, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding gathers.
 *   out[r, :] = sum_t table_t[idx_t[r], :]   for up to 4 tables of equal width C   (f32)
 * Replaces the node-feature gathers of model.py:193-203 (atom + in-degree + out-degree) and of
 * model_fqandtoyo.py:1259-1264,1288-1298 (POI / category / time-slot / degree / position rows).
 * idx_t: [R] int64 or int32 (idx_dtype); a negative index contributes nothing.
 */
int mobgt_embed_gather_sum(const float* const* tables_host, const void* const* idx_host, int n_tables,
                           float* out, int64_t R, int C, int64_t ld_out, int idx_dtype, void* stream);
/* Backward: d_table_t[idx_t[r], :] += dout[r, :] (f32 atomics); rows with idx == skip_idx_t or < 0 are skipped
 * (nn.Embedding(padding_idx=0) semantics when skip = 0, pass -1 to keep every row). */
int mobgt_embed_scatter_add(float* const* d_tables_host, const void* const* idx_host, const int64_t* skip_idx_host,
                            int n_tables, const float* dout, int64_t R, int C, int64_t ld_dout,
                            int idx_dtype, void* stream);
/* Concatenation instead of the sum (`torch.cat((poi_embed, time_embed), -1)` of FuseEmbeddings,
 * model_fqandtoyo.py:1262-1268, on gathered rows): out[r, off_t : off_t + widths[t]] = table_t[idx_t[r], :] with
 * off_t = widths[0] + .. + widths[t-1] (zeros where idx < 0); widths multiples of 4, sum <= ld_out.  The scatter adds
 * dout's column blocks back into the tables (atomics; rows equal to skip_idx[t] or negative are skipped). */
int mobgt_embed_gather_concat(const float* const* tables, const void* const* indices, const int* widths, int n_tables,
                              float* out, int64_t R, int64_t ld_out, int idx_dtype, void* stream);
int mobgt_embed_scatter_concat(float* const* d_tables, const void* const* indices, const int64_t* skip_idx,
                               const int* widths, int n_tables, const float* dout, int64_t R, int64_t ld_dout,
                               int idx_dtype, void* stream);
/* n <= 8 such gathers (+ at most three folded tables, below) over ONE position list in one launch, each into its own destination: job t copies
 * table_t[idx_t[r], :] (width_t columns, zeros where idx < 0) to buf_t[r, coff_t : coff_t + width_t] (row stride ld_t), or ADDS
 * it there when accum_t == 1 (jobs run in order: a sum of tables is a copy followed by adds); accum_t == 2 (round 4): table_t
 * is FOLDED into the last job in front of it that is not itself folded -- same width, read at that job's row index, added in
 * registers before the job stores (forward only, one job per call may have folded tables; the partial tables of
 * mobgt_mask_rows_fwd) -- `[poi ; time]`, the category rows
 * and the additive degree / frequency / positional rows of model_fqandtoyo.py:1259-1298.  backward != 0: buf_t is the
 * gradient buffer, scatter-added into d_table_t (f32 atomics; rows equal to skip_t and null d_tables skipped); extra_row0
 * (optional, [width of job extra_job]) is added to ROW 0 of d_table[extra_job] -- the graph token's share of pe[0]'s
 * gradient (model_fqandtoyo.py:1338-1342), so that the positional table has ONE gradient producer. */
int mobgt_embed_gather_multi(int n, const float* const* tables, float* const* d_tables, const void* const* idx,
                             const int64_t* skip, const int* width, const int* coff, const int* accum, float* const* buf,
                             const int64_t* ld, int64_t R, int idx_dtype, int backward, const float* extra_row0, int extra_job,
                             void* stream);
/* Every row index the node-feature gathers of model_fqandtoyo.py:1259-1264 (POI / time-slot / category),
 * :1287-1298 + :348-351 (positional rows 1..n) need, derived from the padded batch in one launch.
 *   x [G,N] POI ids (0 = pad; x_dtype MOBGT_I64 / I32) and time_normal [G,N] f32, both with element strides (g, n);
 *   poi2cat [P+1] int64 (row 0 = pad).
 * idx [8][G*N] int64 (-1 = "no row" in 0..3):
 *   0: POI row  (rows_only ? the position g*N+n in a per-batch table : x-1)       1: (long)(time_normal*48)
 *   2: poi2cat[x]-1      3: n+1 where n+1 <= number of real nodes of graph g      4: max(x-1, 0)      5: zeros
 *   6, 7: in_degree / out_degree [G*N] (contiguous; deg_dtype MOBGT_I64 / I32 / I16) widened to int64 -- left
 *         untouched when in_degree is NULL
 * real [G*N] f32: 1 for real nodes, 0 for padding. */
int mobgt_node_index(const void* x, int x_dtype, int64_t xs_g, int64_t xs_n, const float* time_normal, int64_t ts_g,
                     int64_t ts_n, const int64_t* poi2cat, const void* in_degree, const void* out_degree, int deg_dtype,
                     int64_t* idx, float* real, int G, int N, int rows_only, void* stream);

/* nn.Linear on a handful of rows (G <= 16, K <= 512, K % 4 == 0; f32): y [G,V] = x [G,K] w[V,K]^T + b[V] -- the
 * classifier head out_proj on the graph tokens (model_fqandtoyo.py:1394) -- and its backward: dx [G,K] (or NULL),
 * dw [V,K] and db [V] (or NULL), all OVERWRITTEN.  Each product streams w / writes dw exactly once. */
int mobgt_skinny_linear_fwd(const float* x, const float* w, const float* b, float* y, int G, int K, int V, void* stream);
int mobgt_skinny_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int G, int K,
                            int V, void* stream);
/* dx = dy @ w alone, on the matrix cores (csrc/skinny.hip): dx [G,K] f32 must be ZERO on entry (f32 atomics); K % 16 == 0. */
int mobgt_skinny_linear_dx(const float* dy, const float* w, float* dx, int G, int K, int V, void* stream);
/* Both gradients of the skinny Linear in one launch (csrc/skinny.hip): dx = dy @ w [G,K] (ZERO on entry: f32 atomics;
 * K % 16 == 0), dw = dy^T x [V,K] (overwritten), db = column sums of dy [V] or NULL. */
int mobgt_skinny_linear_bwd_both(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                 int G, int K, int V, void* stream);
/* logits = x w^T + b [G,V] alone on the matrix cores, one pass over w (csrc/skinny.hip; the kernel of mobgt_skinny_linear_gtl
 * without the loss): G <= 16, K % 64 == 0, K <= 448; x, w 16-byte aligned.  The library's M = 16 GEMM takes 9-29 us here. */
int mobgt_skinny_linear_fwd_mfma(const float* x, const float* w, const float* b, float* y, int G, int K, int V, void* stream);
/* The classifier and its loss in one launch (training; model_fqandtoyo.py:1394 + GradientTailLoss :545-550 as called at
 * :1446-1460): logits = x w^T + b [G,V] (stored only when `logits` != NULL), *loss = mean GradientTailLoss(logits, class of row g
 * = targets[g] + target_offset, alpha), dlogits [G,V] = d loss / d logits.  G <= 16, K % 64 == 0, K <= 448; x, w 16-byte
 * aligned.  Same formulas as mobgt_gradient_tail_loss; the loss's partial sums are added in a fixed order. */
int mobgt_skinny_linear_gtl(const float* x, const float* w, const float* b, const int64_t* targets, int64_t target_offset,
                            float* logits, float* dlogits, float* loss, int G, int K, int V, float alpha, void* stream);

/* Rows of a bf16 matrix a [*, ld] gathered and transposed in one pass: out_rows [R, C] = a[rows[j], 0:C] and
 * out_t [C, R] = out_rows^T (the operands of the "rows only" last GCN layer, modelGNN.py:38-44 restricted to the
 * batch's POI rows).  R, C, ld multiples of 8; pointers 16-byte aligned. */
int mobgt_gather_rows_t(const void* a, int64_t ld, const int64_t* rows, void* out_rows, void* out_t, int R, int C,
                        void* stream);

/* Evaluation (model_fqandtoyo.py:48-90 get_acc, :122-131 MRR_metric): for every row of scores [G,V] f32 the number
 * of classes ranked ahead of target[g] (class id, int64): rank[2g] counts strictly larger scores plus equal scores
 * at a LOWER index (a stable descending top-k: "target is in the top k" <=> rank[2g] < k, its position is rank[2g]);
 * rank[2g+1] counts equal scores at a HIGHER index instead (the reference's reversed ascending argsort in
 * MRR_metric).  -1 for a target outside [0, V). */
int mobgt_target_rank(const float* scores, const int64_t* target, int32_t* rank, int64_t G, int64_t V, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused elementwise / normalisation pieces of EncoderLayer.forward between the library GEMMs
 * (graphormer/model.py:479-489, model_fqandtoyo.py:1731-1743) and their backward.  R rows (= G*T tokens),
 * C columns (<= 512).  The residual stream and all statistics are f32; `act_dtype` (MOBGT_F32 / MOBGT_BF16)
 * is the dtype of the GEMM-facing tensors y, z, dz, dy.  Dropout (nn.Dropout on the branch output,
 * model.py:482,487) uses the attention kernels' counter hash keyed by (seed [+ *seed_dev], salt, row, column).
 *
 * fwd:  x1 = x + dropout(y)          (y == NULL: x1 is not written, x1 := x)
 *       z  = LayerNorm(x1; ln_w, ln_b, eps 1e-5) written as act_dtype (z) and/or f32 (z32); ln_w == NULL: skipped
 * bwd:  dx1 = dres + LayerNorm_bwd(dz + dz32)   -> grad of the residual input x
 *       dy  = dropout_bwd(dx1)                  -> grad of the branch output y (dy == NULL: skipped)
 *       dgamma/dbeta [C] += LayerNorm affine grads, dbias [C] += column sums of dy (bias grad of the Linear
 *       that produced y); all three accumulate with f32 atomics -- zero them first.
 */
int mobgt_dropout_add_ln_fwd(const float* x, const void* y, float* x1, const float* ln_w, const float* ln_b,
                             void* z, float* z32, float* mean, float* rstd, int64_t R, int C,
                             float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt,
                             int act_dtype, void* stream);
int mobgt_dropout_add_ln_bwd(const void* dz, const float* dz32, const float* dres, const float* x1,
                             const float* mean, const float* rstd, const float* ln_w, float* dx1, void* dy,
                             float* dgamma, float* dbeta, float* dbias, int64_t R, int C, float dropout_p,
                             uint64_t seed, const uint64_t* seed_dev, uint32_t salt, int act_dtype, void* stream);
/* Exact-erf GELU (nn.GELU, model.py:398): h = gelu(u);  du = dh * gelu'(u) with dbias [C] += colsum(du). */
int mobgt_gelu_fwd(const void* u, void* h, int64_t n, int act_dtype, void* stream);
int mobgt_gelu_bwd_colsum(const void* dh, const void* u, void* du, float* dbias, int64_t R, int C,
                          int act_dtype, void* stream);
/* GradientTailLoss(inputs, targets, alpha) of graphormer/model_fqandtoyo.py:545-550 (beta = k = 1) and its
 * gradient in one pass: logits [G,V] f32, targets [G] int64 -> *loss (f32 scalar, overwritten) and
 * dlogits [G,V] = d(loss)/d(logits).  The class id of row g is targets[g] + target_offset (training_step's
 * `batched_data.y - 1`, model_fqandtoyo.py:1446-1460, without a launch for the subtraction). */
int mobgt_gradient_tail_loss(const float* logits, const int64_t* targets, int64_t target_offset, float* dlogits,
                             float* loss, int64_t G, int64_t V, float alpha, void* stream);
/* The stock variant's encoder input in one launch each way (model.py:193-205; csrc/layer.hip): y [G, N+1, C] with row (g, 0) =
 * graph_token and row (g, 1 + n) = atom[x[g,n]] + indeg[in_degree[g,n]] + outdeg[out_degree[g,n]], then input_dropout with
 * mobgt_dropout's mask for the same (seed, salt) over rows g (N+1) + t.  Indices [G,N]: x of idx_dtype, the two degree tensors
 * of deg_dtype (MOBGT_I64 / I32 / I16 each);
 * a negative or out-of-table index contributes nothing.  C % 4 == 0, tables and y 16-byte aligned.
 * bwd: the table / token gradients ACCUMULATE (f32 atomics: zero them, or pass gradient sinks; NULL = not wanted); row
 * `padding_idx` of every table receives nothing. */
int mobgt_stock_tokens_fwd(const void* x, const void* in_degree, const void* out_degree, int idx_dtype, int deg_dtype, const float* atom,
                           const float* indeg, const float* outdeg, const float* graph_token, float* y, int G, int N, int C,
                           int64_t n_atom, int64_t n_in, int64_t n_out, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                           uint32_t salt, void* stream);
int mobgt_stock_tokens_bwd(const float* dy, const void* x, const void* in_degree, const void* out_degree, int idx_dtype,
                           int deg_dtype, float* d_atom, float* d_indeg, float* d_outdeg, float* d_graph_token, int G, int N, int C,
                           int64_t n_atom, int64_t n_in, int64_t n_out, int64_t padding_idx, float dropout_p, uint64_t seed,
                           const uint64_t* seed_dev, uint32_t salt, void* stream);
/* Round 4: the stock step's three independent front launches as one grid -- mobgt_stock_tokens_fwd (same arguments), a weight
 * pack (mobgt_pack_mfma_b's job arrays; n_pack = 0: none) and the hop table's forward (mobgt_hop_table_fwd's arguments;
 * has_hop = 0: none).  Same results as the three launches; the consumers (mobgt_build_bias, the first encoder layer) follow
 * on the stream. */
int mobgt_stock_front_fwd(const void* x, const void* in_degree, const void* out_degree, int idx_dtype, int deg_dtype, const float* atom,
                          const float* indeg, const float* outdeg, const float* graph_token, float* y, int G, int N,
                          int C, int64_t n_atom, int64_t n_in, int64_t n_out, float dropout_p, uint64_t seed,
                          const uint64_t* seed_dev, uint32_t salt, int n_pack, const void* const* pack_src,
                          void* const* pack_dst, const int* pack_N, const int* pack_K, const int* pack_transposed,
                          int has_hop, const float* edge_encoder, const float* edge_dis_encoder, float* hop_table, int D,
                          int n_edge, int H, int fp16_roundtrip, void* stream);
/* Round 4: mobgt_stock_tokens_bwd (same arguments) and mobgt_hop_table_bwd (H = 8, n_edge <= 2048, d_hop_table and
 * edge_dis_encoder 16-byte aligned) as one grid -- independent of one another, both in front of the optimizer. */
int mobgt_stock_tail_bwd(const float* dy, const void* x, const void* in_degree, const void* out_degree, int idx_dtype,
                         int deg_dtype, float* d_atom, float* d_indeg, float* d_outdeg, float* d_graph_token, int G, int N, int C,
                         int64_t n_atom, int64_t n_in, int64_t n_out, int64_t padding_idx, float dropout_p, uint64_t seed,
                         const uint64_t* seed_dev, uint32_t salt, const float* d_hop_table, const float* edge_encoder,
                         const float* edge_dis_encoder, float* d_edge_encoder, float* d_edge_dis_encoder, int D, int n_edge,
                         int H, int fp16_roundtrip, void* stream);
/* Round 4: dst[i][e] = sum_{k < s[i]} src[i][k * numel[i] + e] for n <= 48 jobs in one launch (f32, numel % 4 == 0, 16-byte
 * aligned): the sums over the library's split-K partial weight gradients ([s, M, N] from torch.bmm over row slices --
 * fused_layer._mm_tn_f32), deferred to the end of the backward pass and written into the gradients' sinks. */
int mobgt_partial_sum_multi(int n, const float* const* src, float* const* dst, const int* s, const int64_t* numel, void* stream);
/* Round 6: the weight gradients of one encoder layer past 4 096 rows in ONE launch -- autograd of the layer's F.linear calls
 * (graphormer/model.py:436-438 linear_q/k/v, :455 output_layer, :393-405 ffn layer1 / layer2; fq: model_fqandtoyo.py:1687-1712):
 *     part[i][s] = G_i[rows of range s]^T X_i[rows of range s]        i < n <= 4 products, s < S row ranges
 * G_i [R, M_i] (row stride ldg[i]) = the gradient of the Linear's output, X_i [R, N_i] (ldx[i]) = its input, both bf16 row-major;
 * part[i]: [S, M_i, N_i] f32, fully overwritten (the sum over s is the weight gradient: mobgt_partial_sum_multi, or any
 * reduction of the caller's); colsum (or colsum[i]) may be NULL, else [M_i] f32 that receives += column sums of G_i (the bias
 * gradient) by atomics -- zero it first.  M_i, N_i, ldg, ldx multiples of 8, pointers 16-byte aligned, S <= ceil(R / 64).
 * Output tiles of 128 x 256 on v_mfma_f32_32x32x16_bf16, operands staged as they lie in memory and read transposed
 * (ds_read_b64_tr_b16).  mobgt_layer_wgrad_big_tiles(M, N): tiles of one product; mobgt_layer_wgrad_big_splits(R, ntiles): the S
 * that gives about one workgroup per compute unit. */
int mobgt_layer_wgrad_big(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                          float* const* part, float* const* colsum, const int* M, const int* N, int64_t R, int S, void* stream);
int mobgt_layer_wgrad_big_tiles(int M, int N);
int mobgt_layer_wgrad_big_splits(int64_t R, int ntiles);
/* final_ln on the graph-token rows (model.py:211-217: the reference normalises every token, then reads row 0 of every graph):
 * y [G,C] = LayerNorm(enc[g,0,:]; ln_w, ln_b, eps), mean / rstd [G] kept for the backward (csrc/layer.hip).  enc [G,T,C] f32
 * contiguous, C <= 1024.
 * bwd: denc [G,T,C] written in full (zero outside the token rows); dgamma / dbeta [C] ACCUMULATE (f32 atomics): zero them. */
int mobgt_token_ln_fwd(const float* enc, const float* ln_w, const float* ln_b, float* y, float* mean, float* rstd, int G, int T,
                       int C, float eps, void* stream);
int mobgt_token_ln_bwd(const float* dy, const float* enc, const float* mean, const float* rstd, const float* ln_w, float* denc,
                       float* dgamma, float* dbeta, int G, int T, int C, void* stream);
/* F.cross_entropy(logits, targets, ignore_index = ignore_index) with mean reduction -- the stock variant's training loss
 * (model.py:218-285; data.py:76 / :98: NLLLoss(ignore_index = 0) on log-softmax outputs) -- and its gradient in one launch
 * (csrc/layer.hip): *loss, dlogits [G,V] (may be NULL).  G <= 4095, V <= 10 240; a target outside [0, V) counts as ignored for
 * the gradient (torch raises).  One launch at a time (a device-global cell carries the mean's sum). */
int mobgt_cross_entropy(const float* logits, const int64_t* targets, int64_t ignore_index, float* dlogits, float* loss, int G,
                        int V, void* stream);
/* nn.Dropout at the model's input/output/positional/GCN sites (model.py:206, model_fqandtoyo.py:358,1347,1364;
 * modelGNN.py:71): y = keep ? x/(1-p) : 0 with the kernels' counter hash keyed by (seed [+ *seed_dev], salt,
 * i / row_len, i % row_len).  The backward is the same call applied to dy. */
int mobgt_dropout(const float* x, float* y, int64_t n, int row_len, float dropout_p, uint64_t seed,
                  const uint64_t* seed_dev, uint32_t salt, void* stream);
/* out [C] (f32) += column sums of g [R,C]: the bias gradient of a Linear layer. */
int mobgt_colsum(const void* g, float* out, int64_t R, int C, int act_dtype, void* stream);
/* Weight / bias gradient of an nn.Linear y = x W^T + b (model.py:388-403, 406-463; what autograd's
 * F.linear backward computes as dy^T x and dy.sum(0)):  dw [M,N] (f32, row pitch ldw) += g^T x and, when db is
 * not null, db [M] += column sums of g, for g [R,M] (row pitch ldg) and x [R,N] (row pitch ldx), both bf16
 * (act_dtype MOBGT_BF16) or both f32 (MOBGT_F32: rounded to bf16 while loading -- bf16 MFMA operands, f32
 * accumulate -- exactly what a cast kernel in front would produce).  M, N, ldg, ldx even; g, x 4-byte (8-byte for
 * f32) aligned.  Split over R across the grid with
 * f32 atomics into dw/db, which the caller zero-initialises (or pre-loads with a gradient to accumulate into). */
int mobgt_linear_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, float* dw, int64_t ldw, float* db,
                       int64_t R, int M, int N, int act_dtype, void* stream);
/* ... with f32 operands and the activation derivative m(.) of mobgt_small_gemm_f32_act applied to g and / or x while they
 * are loaded (g_mask / x_mask: the activation's output, the operand's layout; either may be null).  db_of_x: db is [N] and
 * receives the column sums of the (masked) x instead of g.  g_masked_out (optional, g's layout, f32): the masked g itself,
 * for a data-gradient GEMM that follows. */
int mobgt_linear_wgrad_masked(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* g_mask, const float* x_mask,
                              float m_pos, float m_neg, float m_zero, float* g_masked_out, float* dw, int64_t ldw, float* db,
                              int db_of_x, int64_t R, int M, int N, void* stream);
/* dw [M,N] (ZERO on entry) = g^T x + out_bias[n] on every row: the same split-K kernel used as a skinny product with a
 * bias -- `adj[rows] @ support + b` of the last GraphConvolution (modelGNN.py:38-44) without an add launch. */
int mobgt_linear_wgrad_bias(const void* g, int64_t ldg, const void* x, int64_t ldx, const float* out_bias, float* dw,
                            int64_t ldw, int64_t R, int M, int N, int act_dtype, void* stream);
/* g bf16 [R,M], x f32 [R,N] (rounded to bf16 while loading): dw (ZERO on entry) = g^T x and db_x [N] (zero on entry, or null)
 * += column sums of x: d(support) = adj[rows]^T dout and b.grad = dout.sum(0) of the rows-only GraphConvolution's backward. */
int mobgt_linear_wgrad_mixed(const void* g_bf16, int64_t ldg, const float* x_f32, int64_t ldx, float* dw, int64_t ldw,
                             float* db_x, int64_t R, int M, int N, void* stream);
/* The same for n <= 32 independent Linear layers over the same R rows in ONE launch (host arrays of n entries each;
 * db may be NULL, or hold NULL entries): the weight gradients of all encoder layers of a backward pass. */
int mobgt_linear_wgrad_group(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                             float* const* dw, const int64_t* ldw, float* const* db, int64_t R, const int* M,
                             const int* N, int act_dtype, void* stream);

/* The same for up to 32 problems that differ in everything: problem q contracts R[q] rows of g[q] [R,M[q]] and x[q] [R,N[q]]
 * (bf16 when in_f32[q] == 0, f32 rounded to bf16 while loading when 1); f32 operands may carry activation masks g_mask[q] /
 * x_mask[q] (operand's layout; m(y) = y > 0 ? mask_vals[3q] : (y < 0 ? mask_vals[3q+1] : mask_vals[3q+2]) multiplies the
 * operand: the derivative of dropout(leaky_relu(.)) taken from the activation's output, model_fqandtoyo.py:452-455,
 * modelGNN.py:66-72); db[q] (or NULL) accumulates the column sums of the masked g, or of x when db_of_x[q].  dw[q] and db[q]
 * accumulate (zero them first).  One launch: the leaf weight gradients of a training step issued together. */
int mobgt_linear_wgrad_multi(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                             const float* const* g_mask, const float* const* x_mask, const float* mask_vals,
                             float* const* dw, const int64_t* ldw, float* const* db, const int* db_of_x,
                             const int64_t* R, const int* M, const int* N, const int* in_f32, void* stream);
/* The same launch carrying the hop table's backward (with_hop != 0: the arguments of mobgt_hop_table_bwd follow, H = 8,
 * n_edge <= 256, d_table / edge_dis_encoder 16-byte aligned) as extra workgroups: it needs nothing the group produces, and
 * nothing but the optimizer reads its results.  n may be 0 only with with_hop == 0. */
int mobgt_linear_wgrad_multi_hop(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                                 const float* const* g_mask, const float* const* x_mask, const float* mask_vals, float* const* dw,
                                 const int64_t* ldw, float* const* db, const int* db_of_x, const int64_t* R, const int* M,
                                 const int* N, const int* in_f32, int with_hop, const float* d_table, const float* edge_encoder,
                                 const float* edge_dis_encoder, float* d_edge_encoder, float* d_edge_dis_encoder, int D, int n_edge,
                                 int fp16_roundtrip, void* stream);

/* End of an encoder layer's backward in ONE launch (R <= 1024 rows): the weight gradients of mobgt_linear_wgrad_group
 * (same arguments) and, side by side with them, the layer's input gradient  c[gM,gN] (f32) += a[gM,gK] x b_kn[gK,gN]
 * (bf16 operands; mobgt_layer_gemm's MOBGT_GEMM_ADD with b_is_kn = 1, in place) -- `dx = dx1 + dqkv Wqkv`, which needs
 * the same dqkv as the weight gradients and nothing they produce. */
int mobgt_layer_backward_tail(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                              float* const* dw, const int64_t* ldw, float* const* db, int64_t R, const int* M,
                              const int* N, int act_dtype, const void* a, int64_t lda, const void* b_kn, int64_t ldb,
                              float* c, int64_t ldc, int gM, int gN, int gK, void* stream);

/* Token assembly at the encoder input (model_fqandtoyo.py:1287-1298, 348-358, 1338-1347) in one launch:
 *   out[g,0,:] = drop_in(drop_pos(token + pe0));  out[g,1+n,:] = drop_in(drop_pos(nf[g,n,:] * real[g,n] + add[g,n,:]))
 * nf, add [G,N,C], real [G,N], token, pe0 [C], out [G,N+1,C], all f32 (out_bf16: optional bf16 copy of out, the
 * first layer's GEMM operand, or NULL); p_pos / p_in the two dropout probabilities (0 = off), masks from the library's
 * counter hash with the given salts (row numbering g*N+n / g / g*T+t).
 * Backward: d_nf, d_add [G,N,C] overwritten; d_token [C] ACCUMULATED (sum over graphs; zero it first). */
int mobgt_assemble_tokens_fwd(const float* nf, const float* real, const float* add, const float* token, const float* pe0,
                              float* out, void* out_bf16, int G, int N, int C, float p_pos, float p_in, uint64_t seed,
                              const uint64_t* seed_dev, uint32_t salt_nf, uint32_t salt_tok, uint32_t salt_in, void* stream);
/* mobgt_assemble_tokens_fwd plus the FIRST encoder layer's QKV projection (qkv [G*(N+1), 3C] bf16 = rows wqkv^T + bqkv, weight
 * packed by mobgt_pack_mfma_b) in one launch (csrc/chain.hip); out_bf16 is required; C in {192, 256}. */
int mobgt_assemble_tokens_qkv(const float* nf, const float* real, const float* add, const float* token, const float* pe0,
                              float* out, void* out_bf16, const void* wqkv_packed, const void* bqkv, void* qkv, int G, int N, int C,
                              float p_pos, float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf,
                              uint32_t salt_tok, uint32_t salt_in, void* stream);
/* Round 4: everything between the GCN tables and the first layer's attention in one launch (csrc/chain.hip,
 * token_fwd_chain_kernel): the forward of mobgt_embed_gather_multi (model_fqandtoyo.py:1259-1298) with the same job list (n,
 * tables, idx, width, coff, accum incl. the folded tables; `slot` names each job's destination: 0 = pt [G*N, W2], 1 = the
 * trailing columns of x4 [G*N, C], 2 = add [G*N, C]), FuseEmbeddings-2 and -4 (model_fqandtoyo.py:444-456, :1268-1269:
 * x4[:, :W2] = leaky(pt w2^T + b2), nf = leaky(x4 w4^T + b4); F.linear layout, f32, full-f32 MFMA products) and
 * mobgt_assemble_tokens_qkv.  pt, x4, add, nf are OUTPUTS (what the backward pass and the weight gradients read).
 * C = 192 and W2 = 160 only (MOBGT_EBADDIM otherwise): the fq model at hidden_dim 128. */
int mobgt_token_fwd_chain(int n, const float* const* tables, const void* const* idx, const int* width, const int* coff,
                          const int* accum, const int* slot, int idx_dtype, float* pt, float* x4, float* add, float* nf, int W2,
                          const float* w2, const float* b2, float slope2, const float* w4, const float* b4, float slope4,
                          const float* real, const float* token, const float* pe0, float* out, void* out_bf16,
                          const void* wqkv_packed, const void* bqkv, void* qkv, int G, int N, int C, float p_pos, float p_in,
                          uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf, uint32_t salt_tok, uint32_t salt_in,
                          void* stream);
int mobgt_assemble_tokens_bwd(const float* dout, const float* real, float* d_nf, float* d_add, float* d_token, int G, int N,
                              int C, float p_pos, float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf,
                              uint32_t salt_tok, uint32_t salt_in, void* stream);

/* Epilogue of a GraphConvolution inside GCN.forward (modelGNN.py:38-44, 62-71): y = dropout(LeakyReLU_slope(x + bias)),
 * [R,C] f32, C % 4 == 0, bias [C] or NULL, dropout_p 0 = off (mask: the library's counter hash, row r, salt `salt`).
 * Backward: dx overwritten from dy and the saved OUTPUT y; dbias [C] (or NULL) ACCUMULATED (zero it first). */
int mobgt_bias_act_fwd(const float* x, const float* bias, float* y, int64_t R, int C, float slope, float dropout_p,
                       uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream);
/* mobgt_bias_act_fwd that also writes y transposed as bf16 [C][ld_t] (ld_t >= R): the next bitmask adjacency product's
 * operand (mobgt_mask_gemm with x null). */
int mobgt_bias_act_fwd_t(const float* x, const float* bias, float* y, void* y_t_bf16, int64_t ld_t, int64_t R, int C, float slope,
                         float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream);
int mobgt_bias_act_bwd(const float* dy, const float* y, float* dx, float* dbias, int64_t R, int C, float slope,
                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream);

/* Head activation chain on the graph-token rows (model_fqandtoyo.py:1353-1364), u [R,C] f32, C <= 512:
 *   out = dropout(ELU(LayerNorm(LeakyReLU_slope(u)) * ln_w + ln_b))   (mean / rstd [R] saved for the backward)
 * Backward: du [R,C] overwritten; dgamma, dbeta [C] ACCUMULATED (zero them first).  Dropout: the library's counter
 * hash, row r of salt `salt` (identical to mobgt_dropout on the [R,C] tensor). */
int mobgt_head_act_fwd(const float* u, const float* ln_w, const float* ln_b, float* out, float* mean, float* rstd, int R,
                       int C, float eps, float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                       uint32_t salt, void* stream);
int mobgt_head_act_bwd(const float* dout, const float* u, const float* ln_w, const float* ln_b, const float* mean,
                       const float* rstd, float* du, float* dgamma, float* dbeta, int R, int C, float eps, float slope,
                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream);

/* torch.optim.AdamW (defaults of model_fqandtoyo.py:1599-1616) over one flat f32 parameter buffer of n elements, in
 * place, with device-resident learning rate and step counter (step t = *step_dev - step_base >= 1) so that a captured
 * graph advances on replay; optionally refreshes a bf16 copy of the parameters (shadow_bf16, may be NULL).
 * Learning rate: *lr_dev, or -- when `sched` (5 device floats: warmup_updates, tot_updates, peak lr, end lr, step
 * offset) is given -- PolynomialDecayLR of graphormer/lr.py:17-31 with power 1 evaluated at step_count = t + offset
 * inside the kernel (no per-step host write).  All f32 pointers 16-byte aligned. */
int mobgt_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n,
                     const float* lr_dev, const float* sched, const int64_t* step_dev, int64_t step_base, float beta1,
                     float beta2, float eps, float weight_decay, void* stream);

/* Small f32 GEMMs of the GCN / FuseEmbeddings / head path (graphormer/modelGNN.py:38-44; model_fqandtoyo.py:444-456 and
 * the data gradients autograd derives from them): c[M,N] = a[M,K] x B (+ bias[N]), all f32, full-f32 products on the
 * matrix core, one wave per 16-row output tile.  b_is_nk = 0: b is [K,N] (x @ W, adj @ support, g @ W);
 * b_is_nk = 1: b is [N,K] (F.linear's weight, g @ W^T).  Any M, N, K >= 1 and any row strides (elements).
 * c_dtype MOBGT_F32, or MOBGT_BF16: the result rounded to bf16 on the way out (`support` as the bf16 adjacency
 * product's operand -- no cast launch). */
int mobgt_small_gemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int b_is_nk, const float* bias, void* c,
                         int64_t ldc, int c_dtype, int M, int N, int K, void* stream);
/* The same product with what surrounds it in GraphConvolution / FuseEmbeddings (modelGNN.py:66-72, model_fqandtoyo.py:452-455):
 *   epilogue (leaky != 0):   c = dropout(leaky_relu(acc + bias, slope))  -- dropout mask as mobgt_bias_act_fwd (salt);
 *   prologue (a_mask given): a[r][k] *= m(a_mask[r][k]),  m(y) = y > 0 ? m_pos : (y < 0 ? m_neg : m_zero)
 *                            -- the derivative of that activation, taken from its OUTPUT y, applied to an incoming
 *                            gradient while it is loaded (a_mask has a's layout and lda);
 *   c_t_bf16 (optional):     the result times c_t_scale[row] (or 1) also -- or only, with c null -- as bf16 TRANSPOSED
 *                            [N][ld_t]: the operand layout of mobgt_mask_gemm, which then needs no transpose launch;
 *   k_b (0 = K):             b as [K,N] has only k_b <= K rows: a is zero-padded to a whole number of 16-deep k-steps
 *                            (16-byte operand loads for a 303-wide input). */
int mobgt_small_gemm_f32_act(const float* a, int64_t lda, const float* a_mask, float m_pos, float m_neg, float m_zero,
                             const float* b, int64_t ldb, int b_is_nk, const float* bias, int leaky, float slope,
                             float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* c, int64_t ldc,
                             int c_dtype, void* c_t_bf16, int64_t ld_t, const float* c_t_scale, int M, int N, int K, int k_b, void* stream);

/* GraphConvolution's adjacency product `torch.spmm(adj, support)` (graphormer/modelGNN.py:38-44) for a normalised
 * adjacency held as CSR (csrc/spmm.hip) -- the form that exists at P = 100 000 POIs (BASELINE configs[4]).
 *   mobgt_spmm_csr        out[i,:] = bias + sum_e val[e] b[col[e],:]  over the entries e of row (rows ? rows[i] : i);
 *                         rowptr int64 [n+1], col int32, val f32; b [n_cols, C] f32 (ldb), out [R, C] f32 (ld_out), C % 4 == 0.
 *   mobgt_spmm_csr_t_rows db[col[e],:] += val[e] g[i,:]  (autograd of the row-subset product; db zero-initialised, atomics).
 *   mobgt_spmm_csr_t_rows_gather  the same gradient WITHOUT atomics, from the CSR of the TRANSPOSE (t_*): db [P, C] is
 *                         written in full (no zero-fill needed); `rows` [R] int64 may repeat a row.  head int32 [P] must be
 *                         all -1 on entry and is all -1 again on return (stream order); nxt int32 [R] is scratch.
 *                         C % 4 == 0, C <= 512.  Three launches: thread the subset into per-row lists, gather, unthread.
 */
/* Everything of an fq encoder layer that is row-local, in ONE launch (csrc/chain.hip; graphormer/model.py:455, :388-403,
 * model_fqandtoyo.py:1731-1743):  y = a wo^T + bo;  x1 = x + dropout(y);  z = ffn_norm1(x1);  u = z w1^T + b1;
 * h = gelu(u);  f = h w2^T + b2;  x2 = x1 + dropout(f);  out = ffn_norm2(x2);  qkv_next = out wq_next^T + bq_next.
 * a [R,C] bf16; x [R,C] f32; weights bf16 [out,in] PACKED by mobgt_pack_mfma_b; biases bf16; LayerNorm weights f32.
 * Written: x1, x2, out f32 [R,C]; z, out_a (= bf16(out)) [R,C], u, h [R,F], qkv_next [R,3C] bf16; mean / rstd [R] f32 of
 * both norms.  wq_next / bq_next / qkv_next null for the last layer.  Dropout masks: those of mobgt_dropout_add_ln_fwd
 * with salt1 / salt2.  (C, F) in {(128, 1024), (192, 1024), (256, 1024)}.  ws: see mobgt_chain_ws_bytes (may be null).
 * R <= 4096: 16-row blocks on clusters of workgroups (one workgroup per block without ws); R > 4096: 64-row workgroups
 * (layer_chain_fwd_big_kernel), same results up to the f32 summation order of h w2^T (three parts). */
/* bf16 weight [N,K] row-major -> MFMA operand order (chain.hip): the 16 bytes W[16g + j][32s + 8q .. +7] go to byte offset
 * ((g K/32 + s) 64 + j + 16q) * 16, so that a wave's B-operand load is one contiguous KB.  n <= 96 jobs in one launch;
 * N % 16 == 0, K % 32 == 0.  transposed[i] != 0: src is [K,N] row-major and its TRANSPOSE is packed (the operand of
 * dX = dY W); transposed may be null. */
int mobgt_pack_mfma_b(int n, const void* const* src, void* const* dst, const int* N, const int* K, const int* transposed,
                      void* stream);
int mobgt_layer_chain_fwd(const void* a, const float* x, const void* wo, const void* bo, const float* n1w, const float* n1b,
                          const void* w1, const void* b1, const void* w2, const void* b2, const float* nxw, const float* nxb,
                          const void* wq_next, const void* bq_next, float* x1, void* z, void* u, void* h, float* x2, float* out,
                          void* out_a, void* qkv_next, float* mean1, float* rstd1, float* mean2, float* rstd2, int64_t R, int C,
                          int F, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2,
                          void* ws, void* stream);
/* The same chain backwards, from d(out) to the gradient of the attention output (csrc/chain.hip):
 *   dx2 = ffn_norm2'(dout);  df = dropout'(dx2);  du = (df w2) * gelu'(u);  dz = du w1;  dx1 = dx2 + ffn_norm1'(dz);
 *   dy = dropout'(dx1);  da = dy wo.
 * w2t / w1t / wot: the weights' transposes packed by mobgt_pack_mfma_b(transposed = 1).  Written: df, dy, da [R,C], du [R,F]
 * bf16; dx1 [R,C] f32.  ACCUMULATED (f32 atomics, zero them first): dnxw, dnxb, db2 (= column sums of df), dn1w, dn1b,
 * dbo (= column sums of dy), [C] each.
 * What the layer ABOVE may leave to this launch (all optional): tail_dqkv [R,3C] bf16 + tail_wqkv_t (its Wqkv^T, packed): `dout`
 * then holds only that layer's dx1 and dout + dqkv Wqkv is formed here, per row block; wg_*: n_wg <= 4 weight-gradient
 * problems dW [M,N] += g^T x (+ db [M] += column sums of g) over the same R rows, bf16 operands as mobgt_linear_wgrad, run
 * by extra workgroups of this launch on the compute units its 16-row blocks leave idle.
 * R > 4096: the 64-row form (layer_chain_bwd_big_kernel) -- post-LN layers only, and through THIS entry point without guests:
 * tail_dqkv / tail_wqkv_t null and n_wg == 0, else MOBGT_EBADDIM (mobgt_layer_chain_bwd_big hosts a tail); b1's gradient is then
 * the caller's mobgt_colsum of du. */
int mobgt_layer_chain_bwd(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                          const float* rstd1, const float* mean2, const float* rstd2, const float* n1w, const float* nxw,
                          const void* w2t, const void* w1t, const void* wot, void* df, void* du, void* dy, void* da, float* dx1,
                          float* dnxw, float* dnxb, float* db2, float* dn1w, float* dn1b, float* dbo, int64_t R, int C, int F,
                          float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2,
                          const void* tail_dqkv, const void* tail_wqkv_t, int n_wg, const void* const* wg_g,
                          const int64_t* wg_ldg, const void* const* wg_x, const int64_t* wg_ldx, float* const* wg_dw,
                          const int64_t* wg_ldw, float* const* wg_db, const int* wg_M, const int* wg_N, void* ws, void* stream);
/* The 64-row form of mobgt_layer_chain_bwd on its own entry point (what that call dispatches to past 4 096 rows, for R of any
 * size): the same arguments without the weight-gradient passengers and the workspace, plus db1 [F] f32 -- b1's gradient, the
 * column sums of du, ACCUMULATED like the other small gradients (null: not formed).  tail_dqkv / tail_wqkv_t as there (the layer
 * above's input gradient dout + dqkv Wqkv finished in front of the first norm; both null: none).  Row tensors 16-byte aligned. */
int mobgt_layer_chain_bwd_big(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                              const float* rstd1, const float* mean2, const float* rstd2, const float* n1w, const float* nxw,
                              const void* w2t, const void* w1t, const void* wot, void* df, void* du, void* dy, void* da,
                              float* dx1, float* dnxw, float* dnxb, float* db2, float* dn1w, float* dn1b, float* dbo, float* db1,
                              int64_t R, int C, int F, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt1,
                              uint32_t salt2, const void* tail_dqkv, const void* tail_wqkv_t, void* stream);
/* The two chain launches for PRE-LN layers -- graphormer/model.py:463-489, the EncoderLayer BASELINE.json's north_star names:
 *     y = self_attention_norm(x); y = attention(y); x = x + dropout(y); y = ffn_norm(x); y = ffn(y); x = x + dropout(y)
 * (round 4; (C, F) = (128, 1024) is instantiated for it).  Forward = mobgt_layer_chain_fwd with
 *     n1w / n1b = this layer's ffn_norm;  nxw / nxb = the NEXT layer's self_attention_norm, wq_next / bq_next its QKV projection
 *     (all four null: no successor in the chain -- then x2 is the only output of the second half);  out = null: the
 *     residual stream that leaves the layer is x2, out_a / qkv_next / mean2 / rstd2 belong to the next layer's norm.
 * Backward = mobgt_layer_chain_bwd_preln, same arguments as mobgt_layer_chain_bwd with
 *     dx2 = dout + nxw-norm'(tail_dqkv tail_wqkv)      (post-LN: dx2 = norm'(dout + tail product))
 *     dnxw / dnxb: the NEXT layer's self_attention_norm gradients;  nxw null: dx2 = dout, dnxw / dnxb / mean2 / rstd2 unused. */
int mobgt_layer_chain_bwd_preln(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                          const float* rstd1, const float* mean2, const float* rstd2, const float* n1w, const float* nxw,
                          const void* w2t, const void* w1t, const void* wot, void* df, void* du, void* dy, void* da, float* dx1,
                          float* dnxw, float* dnxb, float* db2, float* dn1w, float* dn1b, float* dbo, int64_t R, int C, int F,
                          float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2,
                          const void* tail_dqkv, const void* tail_wqkv_t, int n_wg, const void* const* wg_g,
                          const int64_t* wg_ldg, const void* const* wg_x, const int64_t* wg_ldx, float* const* wg_dw,
                          const int64_t* wg_ldw, float* const* wg_db, const int* wg_M, const int* wg_N, void* ws, void* stream);
/* `ws` of the two chain launches: a device buffer of mobgt_chain_ws_bytes() bytes, 16-byte aligned, ZEROED ONCE when it is
 * allocated and then left alone, used by ONE stream at a time.  With it, batches of R <= 16 * (compute units / 2) rows run
 * the CLUSTER form: a 16-row block is shared by 4 (R <= 16 * CUs / 4) or 2 workgroups that each stream a quarter / half of
 * the layer's two big weights and meet once per launch through `ws` (flag-carrying 8-byte words, generation numbers that
 * never need resetting).  Same results as the one-workgroup form up to the f32 summation order of h w2^T and du w1 (NCL
 * partial sums).  ws = null: always the one-workgroup form.  MOBGT_CHAIN_NCL=1|2|4 caps the cluster size. */
int64_t mobgt_chain_ws_bytes(void);
/* Peer waits that give up (round 4; they used to trap, which kills the process -- under data parallelism the rank): a
 * workgroup of the cluster form that does not see its partners' packets within the poll limit counts itself in the 32-bit word at
 * byte offset mobgt_chain_ws_fault_offset() of `ws` and carries on (its results are garbage); the host reads / clears that word
 * between steps (mobgt_amd/train.py: TrainStep.check_faults re-runs the step in the one-workgroup form).  The 32-bit word at
 * mobgt_chain_ws_limit_offset() holds the poll limit in rounds (0 = the default, seconds): a test hook.  No reference
 * counterpart: the reference (graphormer/model_fqandtoyo.py:1731-1743) runs one kernel per op. */
int64_t mobgt_chain_ws_fault_offset(void);
int64_t mobgt_chain_ws_limit_offset(void);
/* Backward of the encoder input from d(tokens) down to the gathered rows in ONE launch (csrc/tokbwd.hip;
 * model_fqandtoyo.py:1264-1298, 1338-1347, FuseEmbeddings 444-456) = mobgt_assemble_tokens_bwd + the two data-gradient GEMMs of
 * FuseEmbeddings-4 / -2 with their LeakyReLU derivatives:  d = dropouts'(dout[g, 1 + n]);  d_add = d;  d_nf = d * real;
 * dx4 = (d_nf * leaky'(y4)) w4;  d_pt = (dx4[:, :W2] * leaky'(y2)) w2;  d_token += sum of the graph-token rows' d.
 * y4 [G*N, C] / y2 [G*N, W2] (row stride ld_y2): the two activations' OUTPUTS; w4 [C,C], w2 [W2,W2] row-major (out, in).
 * Written: d_nf, d_add [G*N, C], dx4 [G*N, C] (row stride ld_dx4), d_pt [G*N, W2]; d_token [C] accumulated (zero it first).
 * Masks / salts as mobgt_assemble_tokens_bwd.  (C, W2) = (192, 160). */
int mobgt_token_bwd_chain(const float* dout, const float* real, const float* y4, const float* y2, int64_t ld_y2, const float* w4,
                          const float* w2, float* d_nf, float* d_add, float* dx4, int64_t ld_dx4, float* d_pt, float* d_token,
                          int G, int N, int C, int W2, float slope4, float slope2, float p_pos, float p_in, uint64_t seed,
                          const uint64_t* seed_dev, uint32_t salt_nf, uint32_t salt_tok, uint32_t salt_in, void* stream);
/* The classifier head in front of out_proj, one launch each way (csrc/head.hip; model_fqandtoyo.py:1239-1240, 1353-1364,
 * FuseEmbeddings 452-455):  x3 = [enc[g, 0, :] | table[user[g] + user_offset]] [G, C+U];  u3 = x3 w3^T + b3 (f32);
 * out = dropout(ELU(LayerNorm(LeakyReLU_slope(u3)))).  Replaces mobgt_head_input_fwd + a small GEMM + mobgt_head_act_fwd (same
 * values: full-f32 products, the dropout mask of mobgt_head_act_fwd with `salt`).  Written: x3, u3, out [G, C+U], mean / rstd [G].
 * C + U in {320, 384}, C and U multiples of 16, G <= 160; user int32 / int64 (MOBGT_I32 / MOBGT_I64); rows of `table` outside
 * [0, n_rows) read as zero.  ws: mobgt_head_chain_ws_bytes() bytes of device memory, 16-byte aligned, ZEROED ONCE at allocation,
 * one stream at a time (a 16-row block is shared by (C + U) / 16 workgroups that exchange their tiles of u3 through it).
 * Backward: dout = d(out) -> du3 [G, C+U] (written: the weight gradient du3^T x3 and the bias gradient are the caller's),
 * denc [G,T,C] WRITTEN IN FULL (token rows = dx3[:, :C], all other rows zero), dtable += dx3[:, C:] at the users' rows,
 * dgamma / dbeta += LayerNorm's (f32 atomics: zero them first). */
int mobgt_head_chain_fwd(const float* enc, const void* user, int user_dtype, int64_t user_offset, const float* table,
                         int64_t n_rows, const float* w3, const float* b3, const float* ln_w, const float* ln_b, float* x3,
                         float* u3, float* out, float* mean, float* rstd, int G, int T, int C, int U, float eps, float slope,
                         float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* ws, void* stream);
int64_t mobgt_head_chain_ws_bytes(void);
/* fault count / poll limit words of that workspace: as mobgt_chain_ws_fault_offset / _limit_offset above */
int64_t mobgt_head_chain_ws_fault_offset(void);
int64_t mobgt_head_chain_ws_limit_offset(void);
int mobgt_head_chain_bwd(const float* dout, const float* u3, const float* mean, const float* rstd, const void* user,
                         int user_dtype, int64_t user_offset, int64_t n_rows, const float* w3, const float* ln_w,
                         const float* ln_b, float* du3, float* denc, float* dtable, float* dgamma, float* dbeta, int G, int T,
                         int C, int U, float eps, float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                         uint32_t salt, void* stream);
/* The whole 3-layer GCN of a SMALL dense graph (graphormer/modelGNN.py:53-74 on the ~300-node category graph,
 * model_fqandtoyo.py:1237) as ONE launch each way (csrc/smallgcn.hip): ceil(n/16) co-resident workgroups that meet at
 * `counter` (int[1], ZERO on entry) between the layers.
 *   forward   h1 = leaky(ax w0 + b0);  t = a h1;  h2 = dropout(leaky(t w1 + b1));  t2 = a h2;  out = t2 w2 + b2
 *             ax = a @ x precomputed [n,K0]; a [n,n]; w* [in,out] row-major; h1/t [n,H1], h2/t2 [n,H2] are kept for the
 *             backward; out [n,H3].  (H1, H2, H3) = (16, 64, 32) (the widths MobGT uses; others: MOBGT_EBADDIM); n <= 4096.  Dropout mask / slope as mobgt_bias_act_fwd.
 *   backward  g = d(out); a_t = a^T [n,n]; dw* / db* are ACCUMULATED (f32 atomics: zero them first); dt2 [n,H2] and
 *             dt [n,H1] are scratch. */
int mobgt_small_gcn_fwd(const float* ax, const float* a, const float* w0, const float* b0, const float* w1, const float* b1,
                        const float* w2, const float* b2, float* h1, float* t, float* h2, float* t2, float* out, int* counter,
                        int n, int K0, int H1, int H2, int H3, float slope, float dropout_p, uint64_t seed,
                        const uint64_t* seed_dev, uint32_t salt, void* stream);
/* The same launch carrying, as passenger workgroups on the compute units the network leaves idle,
 *   - the step's weight pack (mobgt_pack_mfma_b's jobs, same arguments, pack_n <= 96; pack_n = 0: none),
 *   - the index derivation of the node features (with_node_index != 0: the arguments of mobgt_node_index follow) and
 *   - the hop table's forward (with_hop != 0: those of mobgt_hop_table_fwd) -- 4.8 us each as launches of their own, all ramp.
 * Results identical to the separate launches (the pack is 11.7 us of pure data movement beside a 26 us launch that keeps 19
 * compute units busy).  Nothing in this launch may read what a passenger writes.  Not re-entrant (one host thread at a time). */
int mobgt_small_gcn_fwd_pack(const float* ax, const float* a, const float* w0, const float* b0, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* h1, float* t, float* h2, float* t2, float* out, int* counter,
                             int n, int K0, int H1, int H2, int H3, float slope, float dropout_p, uint64_t seed,
                             const uint64_t* seed_dev, uint32_t salt, int pack_n, const void* const* pack_src, void* const* pack_dst,
                             const int* pack_N, const int* pack_K, const int* pack_transposed, int with_node_index, const void* ni_x,
                             int ni_x_dtype, int64_t ni_xs_g, int64_t ni_xs_n, const float* ni_time_normal, int64_t ni_ts_g,
                             int64_t ni_ts_n, const int64_t* ni_poi2cat, const void* ni_in_degree, const void* ni_out_degree,
                             int ni_deg_dtype, int64_t* ni_idx, float* ni_real, int ni_G, int ni_N, int ni_rows_only, int with_hop,
                             const float* hop_edge_encoder, const float* hop_edge_dis_encoder, float* hop_out, int hop_D,
                             int hop_n_edge, int hop_H, int hop_fp16_roundtrip, void* stream);
int mobgt_small_gcn_bwd(const float* g, const float* ax, const float* a_t, const float* w1, const float* w2, const float* h1,
                        const float* t, const float* h2, const float* t2, float* dw0, float* db0, float* dw1, float* db1,
                        float* dw2, float* db2, float* dt2, float* dt, int* counter, int n, int K0, int H1, int H2, int H3,
                        float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream);
/* The same launch carrying mobgt_build_bias_bwd (with_bias != 0: its arguments follow, same meaning) as passenger workgroups
 * on the compute units the network leaves idle.  Covers the short-batch instantiation only -- idx_dtype MOBGT_I16, edge_dtype
 * MOBGT_U8, H = 8, F = 1, 0 < D <= 20, G (N+1)^2 < 2^20 -- else MOBGT_EBADDIM (callers then use the two launches).  Results as
 * the two launches (f32 atomics in another order). */
int mobgt_small_gcn_bwd_bias(const float* g, const float* ax, const float* a_t, const float* w1, const float* w2, const float* h1,
                             const float* t, const float* h2, const float* t2, float* dw0, float* db0, float* dw1, float* db1,
                             float* dw2, float* db2, float* dt2, float* dt, int* counter, int n, int K0, int H1, int H2, int H3,
                             float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, int with_bias,
                             const void* dbias, int dbias_dtype, int n_slices, int64_t slice_stride, const float* attn_bias,
                             const void* rel_pos, const void* poi_pos, const void* edge_input, float* d_rel_table,
                             float* d_poi_table, float* d_hop_table, float* d_vdist, int G, int N, int H, int D_in, int D, int F,
                             int n_rel, int n_poi, int n_edge, int64_t ld_bias, int idx_dtype, int edge_dtype, void* stream);
/* The same adjacency product from a BITMASK of the adjacency (csrc/maskgemm.hip) -- for the reference's
 * (D+I)^-1 (A+I) with a 0/1 matrix A, whose non-zeros of row i all equal 1/(deg_i + 1) (model_fqandtoyo.py:481-486):
 *   out[i,:] = rscale[i] * sum_k bit(i,k) * bscale[k] * x[k,:] + bias          (rscale / bscale / bias may be null)
 * mask: [M, ld_mask_words] uint32 (ld_mask_words % 4 == 0 and * 32 >= roundup(K, 128)), bit (k & 31) of word (k >> 5) of
 * row i = entry (i,k), bits >= K zero; x [K,N] f32 (rounded
 * to bf16 as MFMA operand, f32 accumulate), out [M,N] f32, N in {16, 32, 48, 64}.
 * work: mobgt_mask_gemm_workspace_bytes(K, N) bytes of device scratch (the operand transposed to bf16 [N][K], written by
 * a first small launch so that the product reads 16 bytes per MFMA operand).  x null: `work` ALREADY holds that operand
 * (bf16 [N][roundup(K,128)], times bscale, zero beyond K), written by the kernel that produced x. */
int64_t mobgt_mask_gemm_workspace_bytes(int K, int N);
int mobgt_mask_gemm(const uint32_t* mask, int64_t ld_mask_words, const float* x, int64_t ldx, const float* bscale,
                    const float* rscale, const float* bias, float* out, int64_t ld_out, void* work, int M, int K, int N,
                    void* stream);
/* Round 4 -- a tall-and-narrow small GEMM riding in the bias assembly's launch (csrc/bias.hip):
 *   c = leaky_relu(a [M,K] @ b [k_b <= K rows, N] + bias)  (+ the result transposed in bf16, as mobgt_small_gemm_f32_act writes it)
 * is left as a job for the NEXT short-batch mobgt_build_bias launch of this process (H = 8, G (N+1)^2 < 2^20), which runs it in
 * its split-K form as extra workgroups of the same grid -- the distance GCN's first layer (modelGNN.py:38-44 / 66-72 on the
 * precomputed A X) depends on parameters only and is as independent of the bias assembly as two launches can be.  Shapes of the
 * split-K form only: N <= 16, K % 16 == 0, 128 <= K <= 512, a's rows 16-byte aligned, M >= 1024 (MOBGT_EBADDIM otherwise:
 * launch mobgt_small_gemm_f32_act instead).  a == NULL drops a pending job; mobgt_front_sgemm_pending() != 0: no launch has
 * taken the job yet.  Host state of the library: one thread. */
int mobgt_front_sgemm_job(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, int leaky, float slope,
                          float* c, int64_t ldc, void* c_t_bf16, int64_t ld_t, int M, int N, int K, int k_b);
int mobgt_front_sgemm_pending(void);
/* Round 4 -- the distance GCN's hidden and last layer around two bitmask products, rows-only form (csrc/maskgemm.hip;
 * reference: graphormer/modelGNN.py:38-44 GraphConvolution, :66-72 GCN.forward; model_fqandtoyo.py:1236 the distance GCN over
 * all P POIs, :1264 its table read at the batch's POI ids only).  Hidden widths 16 and 64 (model_fqandtoyo.py: nhid = [16, 64]).
 *   mobgt_mask_gemm_l1_fwd   t [M,16] = rscale * (A y0)   (y0t: y0 transposed, bf16 [16][ld_y0t], zero beyond K);
 *                            y [M,64] = dropout(leaky_relu(t w1 + b1))  (w1 [16,64], mask rule of mobgt_small_gemm_f32_act),
 *                            yt = y transposed, bf16 [64][ld_yt] (columns >= M untouched); `zero`: zero_floats (% 4 == 0) f32
 *                            zeroed as a side job (the next call's atomic destinations; may be null);
 *   mobgt_mask_rows_fwd      u [R,64] = rscale[rows[r]] * (A[rows[r], :] y)  and  parts [4][R,NO] with
 *                            parts[q] = u[:, 16q : 16q + 16] w2[16q : 16q + 16, :] (+ b2 for q = 0): the last layer's output is
 *                            parts[0] + parts[1] + parts[2] + parts[3], added by the consumer (mobgt_embed_gather_multi's
 *                            accumulate jobs) in that order -- no atomics, bit-reproducible;
 *                            rs_rows [R] = rscale[rows[r]] (may be null);  w2 [64,NO], NO % 4 == 0, NO <= 192;
 *   mobgt_mask_rows_bwd      dy1 [P,64] = A[rows,:]^T gu  with  gut = gu^T in bf16 [64][roundup(R,128)] (already scaled by
 *                            rs_rows, zero beyond R);  mask_t: bitmask of A^T;  and the hidden layer's data gradient
 *                            dtt [16][ld_dtt] = (((dy1 * m(y1)) w1^T) * bscale)^T in bf16 -- the operand of
 *                            mobgt_mask_gemm(mask_t, x = null) -- with m(y) = y > 0 ? m_pos : (y < 0 ? m_neg : m_zero).
 * MOBGT_EBADDIM on other widths / leading dimensions, MOBGT_EALIGN on operands that are not 16-byte aligned. */
int mobgt_mask_gemm_l1_fwd(const uint32_t* mask, int64_t ld_mask_words, const float* rscale, const void* y0t_bf16,
                           int64_t ld_y0t, float* t, const float* w1, const float* b1, float slope, float dropout_p,
                           uint64_t seed, const uint64_t* seed_dev, uint32_t salt, float* y, void* yt_bf16, int64_t ld_yt,
                           void* zero, int64_t zero_floats, int M, int K, void* stream);
int mobgt_mask_rows_fwd(const uint32_t* mask, int64_t ld_mask_words, const int64_t* rows, const float* rscale,
                        const void* y1t_bf16, int64_t ld_y1t, const float* w2, const float* b2, float* u, float* parts,
                        float* rs_rows, int R, int K, int NO, void* stream);
int64_t mobgt_mask_rows_bwd_lds_bytes(int64_t ld_mask_words, int R);
int mobgt_mask_rows_bwd(const uint32_t* mask_t, int64_t ld_mask_words, const int64_t* rows, const void* gut_bf16,
                        int64_t ld_gut, const float* y1, float m_pos, float m_neg, float m_zero, const float* w1,
                        const float* bscale, float* dy1, void* dtt_bf16, int64_t ld_dtt, int R, int P, void* stream);
int mobgt_spmm_csr(const int64_t* rowptr, const int32_t* col, const float* val, const int64_t* rows, const float* b,
                   int64_t ldb, const float* bias, float* out, int64_t ld_out, int64_t R, int C, void* stream);
int mobgt_spmm_csr_t_rows(const int64_t* rowptr, const int32_t* col, const float* val, const int64_t* rows,
                          const float* g, int64_t ldg, float* db, int64_t ld_db, int64_t R, int C, void* stream);
int mobgt_spmm_csr_t_rows_gather(const int64_t* t_rowptr, const int32_t* t_col, const float* t_val, const int64_t* rows,
                                 int* head, int* nxt, const float* g, int64_t ldg, float* db, int64_t ld_db, int64_t P,
                                 int64_t R, int C, void* stream);

/* The encoder layer's small GEMMs with the following elementwise step fused (graphormer/model.py:388-403, 406-463;
 * model_fqandtoyo.py:1641-1712 and the autograd of those F.linear calls): bf16 operands, f32 accumulate.
 *   acc[M,N] = A[M,K] x op(B)        A row-major (lda);  b_is_kn = 0: B is [N,K] (an nn.Linear weight, forward),
 *                                     b_is_kn = 1: B is [K,N] (the same weight seen from dX = dY W)
 *   bias [N] bf16 or NULL is added to acc first.  Then, by `epilogue`:
 *     MOBGT_GEMM_BIAS     C (bf16) = acc
 *     MOBGT_GEMM_GELU     C (bf16) = u = acc,  aux_out (bf16, ldc) = gelu(u)  (exact erf; from the rounded u)
 *     MOBGT_GEMM_GELU_BWD C (bf16) = acc * gelu'(aux_in[m,n])   aux_in = u, bf16 (ldc)
 *     MOBGT_GEMM_ADD      C (f32)  = acc + aux_in[m,n]          aux_in f32 (ldc); may alias C
 * K % 32 == 0, N % 8 == 0, lda / ldc % 8 == 0 (ldb % 8 == 0, or % 2 == 0 when b_is_kn), 16-byte aligned pointers. */
#define MOBGT_GEMM_BIAS 0
#define MOBGT_GEMM_GELU 1
#define MOBGT_GEMM_GELU_BWD 2
#define MOBGT_GEMM_ADD 3
int mobgt_layer_gemm(const void* a, int64_t lda, const void* b, int64_t ldb, int b_is_kn, const void* bias, void* c,
                     int64_t ldc, int epilogue, const void* aux_in, void* aux_out, int M, int N, int K, void* stream);

/* The three `dropout_add_ln` steps of an encoder layer whose result is the A operand of a GEMM over the model width,
 * fused into that GEMM as its prologue (csrc/lngemm.hip; bf16 activations, C <= 256, C % 32 == 0).  Arithmetic, dropout
 * masks and side outputs are those of mobgt_dropout_add_ln_fwd / _bwd followed by mobgt_layer_gemm:
 *   fwd  x1 = x + dropout(y);  z = LayerNorm(x1);  out = z . weight^T + bias   (weight [N, C]; epilogue 0 = bias,
 *        1 = GELU: out = u, aux_out = gelu(u))          -- model.py:482-485 + :397-398 (FFN layer 1);
 *        y == x1 == NULL (round 4): z = LayerNorm(x), no residual -- model.py:480-481, the pre-LN layer's
 *        self_attention_norm in front of its QKV projection
 *   bwd  dx1 = dres + LayerNorm'(dz + dz32);  dy = dropout'(dx1);  out = dy . weight_kn  (weight_kn [C, N]; epilogue 0,
 *        or 2 = out * gelu'(aux_in)); dgamma / dbeta / dbias [C] accumulate (zero them first).
 */
int mobgt_ln_gemm_fwd(const float* x, const void* y, float* x1, const float* ln_w, const float* ln_b, void* z,
                      float* mean, float* rstd, int64_t R, int C, float dropout_p, uint64_t seed,
                      const uint64_t* seed_dev, uint32_t salt, const void* weight, int64_t ldw, const void* bias,
                      void* out, int64_t ld_out, int epilogue, void* aux_out, int N, void* stream);
int mobgt_ln_gemm_bwd(const void* dz, const float* dz32, const float* dres, const float* x1, const float* mean,
                      const float* rstd, const float* ln_w, float* dx1, void* dy, float* dgamma, float* dbeta,
                      float* dbias, int64_t R, int C, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                      uint32_t salt, const void* weight_kn, int64_t ldw, void* out, int64_t ld_out, int epilogue,
                      const void* aux_in, int N, void* stream);

/* Start of a training step (the trainer's `optimizer.zero_grad()` + per-step counter): zero-fills two f32 buffers
 * (element counts multiples of 4, 16-byte aligned; either may be empty) and adds 1 to *counter (may be NULL). */
int mobgt_step_prologue(float* zero_a, int64_t n_a, float* zero_b, int64_t n_b, int64_t* counter, void* stream);
/* ... leaving zero_a[skip_begin, skip_end) untouched: the slice of the flat gradient buffer that a kernel of the backward
 * pass OVERWRITES in full (out_proj's weight gradient: 61 % of the S-FSQ model's gradient bytes need no zeroing). */
int mobgt_step_prologue_skip(float* zero_a, int64_t n_a, int64_t skip_begin, int64_t skip_end, float* zero_b, int64_t n_b,
                             int64_t* counter, void* stream);

/* Input of the classifier head, graph-token rows only (model_fqandtoyo.py:1239-1240 `user_embed_model(user - 1)`,
 * :1353-1358 `embed_fuse_model3(output[p][0], user_embedding[p])`'s concatenation):
 *   x3[g, :] = [ enc[g, 0, 0:C] | table[user[g] + user_offset, 0:U] ]      enc [G,T,C] f32, table [n_rows,U] f32,
 *   user [G] (MOBGT_I64 / I32); an out-of-range user row reads as zeros.
 * bwd: denc [G,T,C] = 0 except denc[g,0,:] = dx3[g,0:C];  dtable[user[g]+user_offset, :] += dx3[g, C:] (atomic; the
 * caller zero-fills dtable). */
int mobgt_head_input_fwd(const float* enc, const void* user, int user_dtype, int64_t user_offset, const float* table,
                         int64_t n_rows, float* x3, int G, int T, int C, int U, void* stream);
int mobgt_head_input_bwd(const float* dx3, const void* user, int user_dtype, int64_t user_offset, float* denc,
                         float* dtable, int64_t n_rows, int G, int T, int C, int U, void* stream);

/* Workgroups of the one-launch GCN kernels (mobgt_small_gcn_fwd / _bwd and their passenger forms; graphormer/modelGNN.py:53-74)
 * that gave up at a grid barrier since the last reset -> *count; reset != 0 clears the count.  Synchronous (a device-symbol copy):
 * call it outside any capture, after the work in question has been waited for.  mobgt_small_gcn_set_wait_limit: the give-up limit
 * in ticks of the 100 MHz clock (<= 0: the default, 2 s) -- a test hook. */
int mobgt_small_gcn_faults(int reset, uint32_t* count);
int mobgt_small_gcn_set_wait_limit(int64_t ticks_100mhz);

/* Diagnostic: `workgroups` x `threads` threads (+ lds_bytes of dynamic LDS each) that hold their compute-unit slots for
 * ticks_100mhz ticks of the 100 MHz wall clock and do nothing -- another stream's persistent kernel (the footprint of RCCL's
 * kernels beside the step under the DDP of README.md:62 / entry.py:141) for the co-residency test of the cluster kernels. */
int mobgt_debug_occupy(int workgroups, int threads, int lds_bytes, int64_t ticks_100mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBGT_HIP_H */
