/* magpo.h -- C ABI of libmagpo_hip.so: the MI355X (gfx950) kernels behind the MAGPO Anakin learner.
 *
 * The reference (liyheng/MAGPO) has no FFI: its hot path is one jitted XLA program behind the Python
 * callable `learn: LearnerFn[GPOLearnerState]` (mava/systems/gpo/anakin/rec_magpo.py:91-530, :635-636).
 * This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md); every entry
 * point names the reference computation it replaces.
 *
 * Conventions
 *   - plain C types only; all pointers are DEVICE pointers unless the name ends in _host;
 *   - fp32 row-major; `ld*` / `*_stride` are element strides; `hipStream_t` is passed as void*;
 *   - stream-ordered and asynchronous; no allocation, no host synchronisation; no setters, no environment variables and no
 *     mutable library state that changes results or buffer sizes: every tuning knob (retention chunk size, GRU MFMA mode and block
 *     rows, A/B kernel variants, acting envs per wave) is a per-call argument.  The only process-level state is the thread-local
 *     error string and memoised device queries (occupancy caps, raised dynamic-LDS limits: idempotent);
 *   - return 0 on success, <0 on error (-1 invalid argument, -2 launch failure); the message is
 *     available from magpo_last_error() (thread-local).
 *   - "slab" outputs are per-workgroup partial sums [grid][width] that the caller reduces with
 *     magpo_reduce_slabs (fixed order => bit-stable results).
 */
#ifndef MAGPO_H
#define MAGPO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* magpo_stream_t; /* hipStream_t */

const char* magpo_last_error(void);
int magpo_abi_version(void);

/* ---- K12 PRNG: jax.random threefry2x32 (rec_magpo.py:135,202,439,642,660,699; decode.py:141) ---- */
int magpo_threefry_split(const uint32_t* key, uint32_t* out, long num, magpo_stream_t stream);
int magpo_threefry_random_bits(const uint32_t* key, uint32_t* out, long num, magpo_stream_t stream);
int magpo_key_split_host(const uint32_t* key_host, int num, uint32_t* out_host);
int magpo_random_bits_host(const uint32_t* key_host, int num, uint32_t* out_host);
/* jax.random.fold_in(key, data) on the host (flax's per-parameter init keys: rec_magpo.py:598-604,623 via magpo_amd/params.py) */
int magpo_key_fold_in_host(const uint32_t* key_host, uint32_t data, uint32_t* out_host);

/* ---- K1 CoordSum env + wrappers (coordsum/env.py:55-139, wrappers/{matrax,observation,auto_reset_wrapper,episode_metrics}.py) ----
 * The step entry points of all three envs write one TimeStep (mava/types.py:45-123 MarlEnv.step): reward [N][A], discount [N][A]
 * (nullable: the MAGPO learner never reads it), done [N] = timestep.last(), the next observation (the reset observation after an
 * auto-reset), observation.step_count, the action mask where the env has one, and extras["episode_metrics"] (m_ep_ret, m_ep_len,
 * m_term [N]).  Env state is updated in place. */
int magpo_coordsum_reset(int* step_count, int* target, int* record, uint32_t* key, uint32_t* metrics_key,
                         float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A, int K, int TLIM,
                         int maxval, const uint32_t* env_keys, float* obs, int* obs_step, magpo_stream_t stream);
int magpo_coordsum_step(int* step_count, int* target, int* record, uint32_t* key, uint32_t* metrics_key,
                        float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A, int K, int TLIM,
                        int maxval, const int* actions, int act_stride, float* reward, float* discount,
                        unsigned char* done, float* obs, int* obs_step, float* m_ep_ret, int* m_ep_len,
                        unsigned char* m_term, int auto_reset, magpo_stream_t stream);

/* ---- Level-Based Foraging env + wrappers (mava/wrappers/jumanji.py:171-220 LbfWrapper with the always-on team reward, AgentID, AutoReset,
 * RecordEpisodeMetrics; the env itself is jumanji LevelBasedForaging-v0 with RandomGenerator(grid_size, fov, num_agents, num_food,
 * max_agent_level, force_coop), configs/env/scenario/*-coop.yaml).  UNPINNED DYNAMICS: Jumanji's source is not part of the reference tree;
 * csrc/lbf.hip and oracle/lbf.py restate its published algorithm and agree bit for bit with each other.
 * State per env: agent_pos [A][2], agent_level [A], food_pos [NF][2], food_level [NF], food_eaten [NF] u8, step_count, key [2], metrics_key [2],
 * episode-metric counters.  obs [N][A][A + 3 (NF + A)] f32 = [one-hot id | (x, y, level) of foods, self, others], mask [N][A][6] u8. */
int magpo_lbf_reset(int* agent_pos, int* agent_level, int* food_pos, int* food_level, unsigned char* food_eaten,
                    int* step_count, uint32_t* key, uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret,
                    int* ep_len, int N, int A, int NF, int G, int fov, int max_level, int force_coop, int time_limit,
                    const uint32_t* env_keys, float* obs, int* obs_step, unsigned char* mask, magpo_stream_t stream);
int magpo_lbf_step(int* agent_pos, int* agent_level, int* food_pos, int* food_level, unsigned char* food_eaten,
                   int* step_count, uint32_t* key, uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret,
                   int* ep_len, int N, int A, int NF, int G, int fov, int max_level, int force_coop, int time_limit,
                   const int* actions, int act_stride, float* reward, float* discount, unsigned char* done, float* obs,
                   int* obs_step, unsigned char* mask, float* m_ep_ret, int* m_ep_len, unsigned char* m_term,
                   int auto_reset, magpo_stream_t stream);

/* ---- Robot Warehouse env + wrappers (mava/wrappers/jumanji.py:137-168 RwareWrapper, AgentID, AutoReset, RecordEpisodeMetrics; the env
 * itself is jumanji RobotWarehouse-v0 with RandomGenerator(column_height, shelf_rows, shelf_columns, num_agents, sensor_range,
 * request_queue_size), configs/env/scenario/tiny-4ag.yaml ...).  UNPINNED DYNAMICS like LBF: csrc/rware.hip and oracle/rware.py restate
 * the published algorithm and agree bit for bit.  State per env: grid_a / grid_s [H][W] (agents / shelves layer, 0 = empty, id + 1),
 * agent_pos [A][2], agent_dir [A], agent_carry [A] u8, shelf_req [NS] u8, queue [Q], step_count, amask [A][5] u8, key [2], metrics_key [2],
 * episode-metric counters (H, W, NS from magpo_rware_layout).  obs rows [N][A] of ldo floats = [one-hot id | 8 + 7 (2 r + 1)^2 features]. */
int magpo_rware_layout(int column_height, int shelf_rows, int shelf_columns, int* out);
int magpo_rware_reset(int* grid_a, int* grid_s, int* agent_pos, int* agent_dir, unsigned char* agent_carry,
                      unsigned char* shelf_req, int* queue, int* step_count, unsigned char* amask, uint32_t* key,
                      uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A,
                      int column_height, int shelf_rows, int shelf_columns, int sensor_range, int queue_size,
                      int time_limit, const uint32_t* env_keys, float* obs, long ldo, int* obs_step,
                      unsigned char* mask, magpo_stream_t stream);
int magpo_rware_step(int* grid_a, int* grid_s, int* agent_pos, int* agent_dir, unsigned char* agent_carry,
                     unsigned char* shelf_req, int* queue, int* step_count, unsigned char* amask, uint32_t* key,
                     uint32_t* metrics_key, float* run_ret, int* run_len, float* ep_ret, int* ep_len, int N, int A,
                     int column_height, int shelf_rows, int shelf_columns, int sensor_range, int queue_size,
                     int time_limit, const int* actions, int act_stride, float* reward, float* discount,
                     unsigned char* done, float* obs, long ldo, int* obs_step, unsigned char* mask, float* m_ep_ret, int* m_ep_len,
                     unsigned char* m_term, int auto_reset, magpo_stream_t stream);

/* input classes of wrapped CoordSum tokens (first-layer tables, csrc/classtab.hip): cls_enc = ((agent * maxval + target) * npos + pos),
 * cls_dec = prev * npos + pos per row; class_rows writes the distinct rows in class order: obs_tab [A*maxval*npos][A+1], pos_enc,
 * and prev_dec / pos_dec [(K+1)*npos].  The actor's class (agent, target) is cls_enc / npos (or cls_enc itself with pos = NULL, npos = 1;
 * prev / cls_dec may then be NULL as well). */
int magpo_coordsum_classes(const float* obs, int F, const int* prev, const int* pos, int A, int maxval, int npos,
                           int* cls_enc, int* cls_dec, long R, magpo_stream_t stream);
int magpo_coordsum_class_rows(int A, int maxval, int npos, int K, float* obs_tab, int* pos_enc, int* prev_dec, int* pos_dec,
                              magpo_stream_t stream);

/* ---- dense layers on fp32 MFMA (flax nn.Dense / retention projections) ----
 * act: 0 none, 1 relu, 2 gelu(tanh), 3 swish, 4 mask: Y = (M > 0) ? XW+b : 0 with the mask M passed in the Ypre argument (same stride as Y)
 * -- the ReLU backward fused into dX = dY W^T.  Ypre (act 0-3, nullable): receives the pre-activation. */
/* variant: 0 = fast path; A/B reference kernels with the same result up to fp32 summation order: linear bit 0 = wave-autonomous kernels
 * instead of the shared-tile ones, bit 1 = the same for KIN = 64 only, bit 2 = KIN 128 / 192 with at least 128 output columns (four-wave column blocks) on bf16 MFMA with both operands
 * split into three bf16 pieces (24 mantissa bits, six products, fp32 accumulate: fp32 accuracy at 6/16 of the fp32 MFMA time; ignored for other shapes); wgrad bit mask 1 = split kernel for every shape, 2 = generic whole-matrix
 * kernel also on full tiles, 4 = no unpadded 64 x 256 kernel, 8 = 128 x 384 as two column halves, 16 / 32 = 64 x 64 / 64 x 256 on the wave-grid kernel,
 * 64 = 128 x 384 on bf16 MFMA with three-piece operand splits (opt-in: faster, but its accumulation error is ~1.2 x the fp32-MFMA kernel's). */
int magpo_linear(const float* X, int ldx, const float* Wt, const float* bias, float* Y, int ldy, float* Ypre,
                 long R, int KIN, int NOUT, int act, int variant, magpo_stream_t stream);
int magpo_linear_pro(int pro, const float* a, long lda, const float* y, long ldy_in, const float* s1, const float* s2,
                     const float* pe, const int* pos, long pos_stride, int npos, int use_pe, const float* W,
                     const int* idx, long idx_stride, const float* s_obs, int F, float* out, long ldout,
                     float* outpe, long ldoutpe, const float* Wt, const float* bias, float* Y, long ldy, long R,
                     int NOUT, magpo_stream_t stream);
long magpo_wgrad_workspace_floats(int KIN, int NOUT, int G);
int magpo_wgrad(const float* X, int ldx, const float* dY, int ldy, long R, int KIN, int krows, int NOUT, float* dW,
                float* db, float* workspace, int G, float scale, int accumulate, int variant, magpo_stream_t stream);
int magpo_reduce_slabs(const float* slab, float* out, int G, long P, long stride, float scale, int accumulate,
                       magpo_stream_t stream);
int magpo_transpose_pad(const float* W, float* Wt, int K, int N, int Npad, magpo_stream_t stream);
int magpo_small_linear(const float* X, int ldx, int F, const float* W, const float* b, float* Y, int ldy, int N,
                       long R, int relu, magpo_stream_t stream);

/* ---- token-local Sable rows (sable_network.py:62-71,93-137,188-217,255-319; retention.py:289-294) ----
 * E = row width = embed_dim of the device network: 64 (16 lanes x float4 per row) or 128 (32 lanes); slabs are [grid][E]. */
int magpo_row_grid(long R);
int magpo_pe_table(float* pe, int npos, int E, magpo_stream_t stream);
/* embed: z (forward) and z / dz (backward) are nullable -- the backward then recomputes the pre-activation from obs / idx (pass W) */
int magpo_embed_fwd(int mode, const float* obs, int ldo, int F, const float* s_obs, const float* W,
                    const int* idx, int idx_stride, const float* s_ln, const float* pe, const int* pos,
                    int pos_stride, int npos, float* z, int ldz, float* xn, int ldxn, float* kin, int ldkin,
                    long R, int E, magpo_stream_t stream);
int magpo_embed_bwd(int mode, const float* z, int ldz, const float* d0, int ldd0, const float* d1, int ldd1,
                    const float* d2, int ldd2, const float* s_ln, float* dz, int lddz, float* slab_sln,
                    float* slab_w, int nrows, const float* obs, int ldo, int F, const float* s_obs, const float* W,
                    float* slab_sobs, const int* idx, int idx_stride, long R, int E, magpo_stream_t stream);
int magpo_small_relu_wgrad(const float* X, int ldx, int F, const float* Yact, const float* dY, float* slab_w, long R,
                           magpo_stream_t stream);
int magpo_small_operand(int mode, const float* obs, int ldo, int F, const float* s_obs, const int* idx,
                        int idx_stride, float* out, long R, magpo_stream_t stream);
int magpo_retpost_fwd(const float* r, int ldr, const float* gp, int ldg, const float* gamma, const float* beta,
                      float* u, int ldu, long R, int hs, int gs, int E, magpo_stream_t stream);
int magpo_retpost_bwd(const float* r, int ldr, const float* gp, int ldg, const float* gamma, const float* beta,
                      const float* du, int lddu, float* dr, int lddr, float* dgp, int lddg, float* slab_gamma,
                      float* slab_beta, long R, int hs, int gs, int E, magpo_stream_t stream);
int magpo_resnorm_fwd(const float* a, int lda, const float* y, int ldy, const float* s1, const float* s2,
                      const float* pe, const int* pos, int pos_stride, int npos, float* out, int ldout,
                      float* outpe, int ldoutpe, long R, int E, magpo_stream_t stream);
int magpo_resnorm_bwd(const float* a, int lda, const float* y, int ldy, const float* s1, const float* s2,
                      const float* d0, int ldd0, const float* d1, int ldd1, const float* d2, int ldd2,
                      float* dsum, int lddsum, float* slab_s1, float* slab_s2, long R, int E, magpo_stream_t stream);
int magpo_headmid_fwd(const float* hpre, int ldh, const float* s, float* hn, int ldhn, const float* w,
                      const float* b, float* value, int value_stride, long R, int E, magpo_stream_t stream);
int magpo_headmid_bwd(const float* hpre, int ldh, const float* s, const float* dhn, int lddhn, const float* w,
                      const float* dvalue, int dvalue_stride, float* dhpre, int lddh, float* slab_s,
                      float* slab_w, float* slab_b, long R, int E, magpo_stream_t stream);
/* wide observations (obs_dim > 32: rows padded to 128 columns, first layers on the MFMA dense kernels; csrc/wideobs.hip):
 * on = RMSNorm_F(obs) * s_obs (sable_network.py:93-95), its s_obs gradient as [magpo_obsnorm_grid(R)][128] slabs, and x + pe[pos] */
int magpo_obsnorm_grid(long R);
int magpo_obsnorm_fwd(const float* obs, long ldo, int F, const float* s_obs, float* on, long R, magpo_stream_t stream);
int magpo_obsnorm_bwd(const float* obs, long ldo, int F, const float* don, float* slab_s, long R, magpo_stream_t stream);
int magpo_add_pe(const float* x, long ldx, const float* pe, const int* pos, long pos_stride, int npos, float* out,
                 long ldout, long R, int E, magpo_stream_t stream);
int magpo_relu_bwd(const float* act, const float* dy, float* dx, long n, magpo_stream_t stream);
int magpo_add_inplace(float* dst, const float* src, long n, magpo_stream_t stream);
/* dst[r][0..W) += src[r][0..W) over R rows (row strides ldd / lds): the partial sums of the blockwise 128-wide retention head */
int magpo_add_rows(float* dst, long ldd, const float* src, long lds, long R, int W, magpo_stream_t stream);

/* ---- K2/K7 retention (retention.py:66-115 chunkwise + recurrent, :117-213 decay matrix / xi) ---- */
/* qkv_rows (nullable): q | k | v are row tables (block-0 projections exist once per distinct input row, csrc/classtab.hip) and token row r
 * reads table row qkv_rows[r]; r, dr, dq, dk, dv are always per token row. */
/* chunk_tokens: tokens per chunk of the chunkwise kernels, 32 (= 0, the default: two workgroups per CU, less masked work; teams of more than 32
 * agents fall back to 64) or 64.  The forward (saved chunk-entry states [nseq][num_chunks][64][64]) and the backward of one pass must be given
 * the same value, and so must magpo_retention_num_chunks when it sizes `states`. */
int magpo_retention_num_chunks(int T, int A, int chunk_tokens);
int magpo_retention_chunk_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                              float* r, long ldr, const float* s0, const int* seq_env,
                              const unsigned char* dones, float* states, float* s_final, int nseq, int T, int A,
                              int masked, float kappa, int hs, const int* qkv_rows, int chunk_tokens, magpo_stream_t stream);
int magpo_retention_chunk_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                              const float* dr, long lddr, float* dq, long lddq, float* dk, long lddk, float* dv,
                              long lddv, const unsigned char* dones, const float* states, int nseq, int T, int A,
                              int masked, float kappa, int hs, const int* qkv_rows, int chunk_tokens, magpo_stream_t stream);
int magpo_retention_recurrent(float* S, const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                              long env_stride_rows, float* r, long ldr, int nenv, int ntok, int ret_from, float decay,
                              int write_state, const float* gp, long ldg, const float* gamma, const float* beta,
                              int hs, int gs, magpo_stream_t stream);
int magpo_zero_states_where_done(float* s0, float* s1, float* s2, const unsigned char* done, int nenv,
                                 magpo_stream_t stream);

/* ---- K2 fused acting step: SableNetwork.get_actions (sable_network.py:443-482; decode.py:111-153) in ONE launch ----
 * dims_host[16] = {N, A, K, F, n_block, n_head, hs, gs, npos, value_only, obs row stride (>= F), envs per wave (0 = by size, or 4 / 8 / 16),
 *   pending, flush, precand, defer}; kappa_host[4] (per head);
 * keys_host [A][2] sampling keys by value, or NULL with ptrs[3] = device key table (static arguments for graph replay);
 * Decoder states are read once and written once per step: the update S <- kappa S + sum_a k_a^T v_a of a step is DEFERRED to the next launch
 *   (the step's k | v rows stay in the scratch rows qkvg1 / kvg2, which therefore must persist between the launches of a rollout).
 *   pending = 1: the previous launch left such rows (apply them first); flush = 1: also apply this launch's rows before returning (with
 *   value_only: the pending ones), so that S_d1 / S_d2 hold the carried states again.  A stand-alone step is {pending 0, flush 1}; a rollout is
 *   {0, 0}, {1, 0} ... and ends with a launch that has flush = 1 (e.g. the bootstrap-value launch {value_only 1, pending 1, flush 1}).
 * defer = 1 (env-step launches of a rollout; excludes flush / value_only): with one head, every other group of workgroups runs the block-0
 *   candidate pre-pass of the NEXT step at the end of this launch (S_d1 of their envs then already holds this step's rows; their candidate
 *   table assumes step count + 1 and no episode end); precand = 1 tells the next launch (same N, same envs per wave) that this happened: those
 *   workgroups skip the pre-pass, zero state and table where `done` says the episode ended, and a flushing launch leaves their S_d1 alone.  A
 *   rollout is {pending, precand, defer, flush} = {0,0,1,0}, {1,1,1,0} ... and ends with the value launch {1,1,0,1}; a stand-alone step is
 *   {0,0,0,1}.  Purpose: these workgroups stream states while the others decode (the launch alternates HBM-bound and dense phases).
 * ptrs_host[49]: obs pos mask keys_dev | s_obs W_obs s_encln W_act s_decln | vh0_t vh0_b vh_s vh_w vh_b1 | h0_t h0_b h_s h1_t h1_b |
 *   pe | S_enc S_d1 S_d2 ([n_block][n_head][N][64][64], updated in place) | scratch xn ([N*A] rows), done [N] u8 or NULL (envs whose
 *   episode just ended: their carried states read as zero, rec_magpo.py:164-169), scratch qkvg u y rep reppe hv ([N*A] rows) |
 *   xa kin1 y1 c cpe y2 xo xope hp hn logits ([N] rows) | u1 u2 ([N*A] rows) | prev [N][A] i32 | action [N][A] i32, logp, value [N][A] |
 *   ptab [N][K + 2][64] scratch (n_head = 1: block-0 self-retention candidate table);
 * blk_ptrs_host[21 * n_block]: qkvg_t wo_t ln1 ln2 gn_g gn_b | qkvg1_t wo1_t dln1 gn1_g gn1_b | q2_t kvg2_t wo2_t dln2 dln3 gn2_g gn2_b |
 *   scratch qkvg1 [N*A][256], q2 [N*A][64], kvg2 [N*A][256] (rows [k | v | - | q2 (kappa S)]).
 *   *_t = the transposed weights of magpo_transpose_pad in the FRAGMENT-MAJOR layout of magpo_act_weight_layout (below): a wave streams
 *   its weight fragments from L2 for every token, and one fragment load must be 1 KB contiguous for that stream to run at the L1 rate. */
int magpo_sable_act(const int* dims_host, const float* kappa_host, const uint32_t* keys_host, const void* const* ptrs_host,
                    int nptrs, const void* const* blk_ptrs_host, int nblk_ptrs, magpo_stream_t stream);
/* Wf[g][gk][lane = m + 16 kq][c] = Wt[16 g + m][16 gk + 4 kq + c] for a transposed weight Wt [nrows][64] (nrows a multiple of 16):
 * the weight operand layout of magpo_sable_act.  Rebuilt by the host after every parameter update (SableGuider.build_act_weights). */
int magpo_act_weight_layout(const float* Wt, float* Wf, int nrows, magpo_stream_t stream);
/* Envs per wave (4, 8 or 16) of the launch magpo_sable_act makes for N envs of A agents; forced = dims_host[11] (0 = chosen by size).
 * The kernel instance is k_sable_act<envs per wave, A <= 4 ? 4 : 8, (n_head == 1 && A <= 4) ? 1 : 0>.  Returns -1 for an invalid `forced`.
 * Shape limits of magpo_sable_act: 1 <= A <= 8, n_block <= 4, n_head in {1, 2, 4}, K <= 31 (the candidate pre-pass holds K + 2 <= 48 rows). */
int magpo_sable_act_envs_per_wave(int N, int A, int forced);

/* ---- fused training-forward segment between two retention ops (sable_network.py:62-71,188-217,277-319; retention.py:289-295; n_head = 1):
 *   u = swish(g) * GroupNorm(r) ; y = u W_o ; o = rms(res + y) s1 [-> rms s2] ; ope = o + pe[pos] ; then the tail
 *   tail 1 (encoder): hv = o W0 + b0, value = rms(gelu(hv)) hs . hw + hb1, q2[k] = ope Wq2[k]   tail 2: out0 = ope W0 (192 columns)
 *   tail 3 (last decoder block): hp = o W0 + b0, hn = rms(gelu(hp)) hs, logits = hn W1 + b1 (K columns)   tail 0: none
 * dims_host[6] = {tail, K, npos, ldg, ld0, nq2}; ptrs_host[34] (device pointers): r gp gamma beta wo_t res s1 s2 pe pos | u y o ope |
 *   w0_t b0 out0 | hs hw hb1 value | q2_t[4] q2[4] | hn w1_t b1 logits | rows   (y, o, ope, s2, rows and unused tail pointers may be NULL;
 *   with rows (i32 [R]) gp and res are row tables and token row r reads table row rows[r]) */
int magpo_seg_post(const int* dims_host, long R, const void* const* ptrs_host, int nptrs, magpo_stream_t stream);

/* backward of that front: dsum = d(res + y) through the RMSNorm(s), du = dsum W_o^T, (dr, dg) through GroupNorm + swish gate, and the
 * parameter-gradient rows (s1, s2, gamma, beta) as [magpo_seg_bwd_grid(R)][64] slabs.  ptrs_host[21]: a y s1 s2 d0 d1 d2 wo_nat r gp gamma beta |
 * dsum dr dgp | slab_s1 slab_s2 slab_ga slab_be | rows | wo_t   (y, s2, d1, d2, slab_s2, rows, wo_t may be NULL; with rows, a and gp are row
 * tables; with wo_t -- W_o^T as magpo_seg_post takes it -- y is not read but recomputed from r and gp, and magpo_seg_post may be given y = NULL) */
int magpo_seg_bwd_grid(long R);
int magpo_seg_bwd(long R, long ldg, long lddg, const void* const* ptrs_host, int nptrs, magpo_stream_t stream);

/* ---- K3/K8 GRU actor (base.py:121-184; flax GRUCell) ----
 * gates [R][512] is an opaque save-for-backward buffer written by magpo_gru_scan_fwd and read by magpo_gru_scan_bwd
 * ((r, z, n, h W_hn + b_hn) interleaved per hidden column).
 * Per-call tuning: split_bf16 -- the TRAINING scans (T > 1 with all save buffers) on 0 = fp32 MFMA; 1 = bf16 pairs (x = hi + lo, product =
 * hi*hi + hi*lo + lo*hi with fp32 accumulation, ~2^-16 relative per product; forward and backward); 2 = bf16 triples (x = hi + mid + lo: the 24
 * mantissa bits of an fp32 operand, six products hh hm mh hl lh mm = 6/16 of the fp32 MFMA time at fp32 accuracy; forward scan, the backward --
 * bound by its gate / gradient traffic -- stays on fp32 MFMA).  Forward and backward of one pass take the same value.  block_rows -- recurrent rows per workgroup of the fp32 scans: 0 = by size (32 when 64-row blocks would leave half of the compute
 * units idle), or 32 / 64 forced.
 * xi_cls (nullable): xi is a table over the distinct input rows and token row r takes xi[xi_cls[r]] (csrc/classtab.hip). */
int magpo_gru_scan_fwd(const float* xi, const float* Wht, const float* b_hn, const float* h0, const int* h0_idx,
                       const unsigned char* reset, float* hs, float* gates, float* hprev, int nseq, int T, int A,
                       const int* xi_cls, int split_bf16, int block_rows, magpo_stream_t stream);
/* hidden-state carry over a TIME-MAJOR trajectory: xi rows (t, env, agent), reset_tm [T][nenv]; writes only the state after step T-1 */
int magpo_gru_carry(const float* xi, const float* Wht, const float* b_hn, const float* h0, const unsigned char* reset_tm,
                    float* h_last, int nenv, int T, int A, const int* xi_cls, int block_rows, magpo_stream_t stream);
/* backward scan.  dg [R][512] = the gate pre-activation gradients of every token row as ONE matrix (dn_in | dr | dz | dn_hid): the
 * gradient of the input projection xi = (r | z | n) is its columns (128..383 | 0..127) and that of the hidden projection h W_h its columns
 * 128..511 (dr and dz are shared by the two sides and written once; ld 512 either way).  slab_bhn [ceil(nseq A / 64)][128]: per-block
 * column sums of dn_hid (the b_hn gradient, folded by magpo_reduce_slabs). */
int magpo_gru_scan_bwd(const float* gates, const float* hprev, const unsigned char* reset, const float* dhs,
                       const float* Wh, float* dg, float* slab_bhn, int nseq, int T, int A,
                       int split_bf16, int block_rows, magpo_stream_t stream);

/* ---- K2 sampling, K5 GAE, K6 shuffle/layout, K9 losses (decode.py:128-149; multistep.py:24-68; rec_magpo.py:222-370,439-462) ---- */
int magpo_sample_categorical(const float* logits, long ld, const unsigned char* mask, long mask_stride,
                             uint32_t k0, uint32_t k1, const uint32_t* key_dev, int* action, long act_stride, float* logp,
                             long logp_stride, int* next_idx, long next_stride, float* lp_all, long lp_ld, int N,
                             int K, magpo_stream_t stream);
/* GAE (multistep.py:24-68): done [T][N] with done[t] = "the observation of step t starts an episode", last_done [N] the same for step T.
 * N * A < 8192 sequences (and T >= 16): a wavefront prefix scan over time, one wave per (env, agent) sequence (log-depth instead of T
 * dependent steps); otherwise one thread per sequence, coalesced across sequences.  Same recurrence, fp32 summation order differs. */
int magpo_gae(const float* reward, const float* value, const unsigned char* done, const float* last_val,
              const unsigned char* last_done, float* adv, float* targets, int T, int N, int A, float gamma,
              float lam, magpo_stream_t stream);
int magpo_gather_minibatch(const float* obs, const int* action, const int* stepcount, const unsigned char* done,
                           const unsigned char* mask, const float* value, const float* logp, const float* adv,
                           const float* targets, const int* env_idx, const int* agent_perm, float* o_obs,
                           int* o_action, int* o_prev, int* o_pos, unsigned char* o_done, unsigned char* o_mask,
                           float* o_value, float* o_logp, float* o_adv, float* o_targets, int* o_h0idx, int T,
                           int N, int A, int F, int K, int mb, magpo_stream_t stream);
/* ---- input-class tables (csrc/classtab.hip): a first layer applied to R rows that take only C << R distinct values is the layer
 * applied to the C distinct rows + a row gather; its parameter gradient is the layer's backward on per-class sums of dY.
 * (mava/networks/base.py:166-170 pre-torso + GRU input projection; sable_network.py:93-101,257-267 obs / action embeddings)
 * class_sum: order [R] = row ids sorted stably by class, offsets [C+1] = class boundaries in order (both int64, device);
 * partial [magpo_class_sum_slots(C)][C][W] workspace; out [C][W].  Bit-stable (data-dependent order only). */
int magpo_gather_rows(const float* table, long ldt, const int* cls, float* out, long ldo, long R, int W,
                      magpo_stream_t stream);
int magpo_class_sum_slots(int C);
int magpo_class_sum(const float* X, long ldx, const long* order, const long* offsets, int C, int W, float* partial,
                    float* out, magpo_stream_t stream);
int magpo_adv_moments(const float* x, long n, double* workspace, float* out, magpo_stream_t stream);
int magpo_loss_fwd_bwd(const float* g_logits, long ldg, const float* a_logits, long lda, const unsigned char* mask,
                       const int* action, const float* old_logp, const float* old_value, const float* value,
                       const float* adv, const float* targets, const float* adv_stats, float* dg_logits,
                       long lddg, float* da_logits, long lddda, float* dvalue, double* workspace, float* loss_out,
                       long R, int K, float clip_eps, float clip_gpo, float ent_coef, float vf_coef, float alpha,
                       magpo_stream_t stream);
int magpo_copy_rows(const float* src, long lds_, float* dst, long ldd, long R, int W, magpo_stream_t stream);

/* ---- K10 optimiser: optax clip_by_global_norm + adam(eps=1e-5) (rec_magpo.py:581-589, :412-420) ---- */
int magpo_clip_adam(float* params, const float* grads, float* mu, float* nu, long n, float grad_scale,
                    float max_norm, float lr, float b1, float b2, float eps, float bc1, float bc2,
                    double* workspace, float* gnorm, magpo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGPO_H */
