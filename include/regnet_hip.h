/*
 * regnet_hip.h -- C ABI of libregnet_hip.so: the MI355X (gfx950) replacement for the native part
 * of REGNet's PointNet++ hot path (the reference's CUDA extensions `pn2_ext` and `dgcnn_ext`).
 *
 * Conventions (all entry points):
 *   - plain device pointers + int64 sizes/element-strides, no torch types, no allocation, no
 *     global state that affects results (the only statics: a per-device "large-LDS attribute set" bit of the two
 *     row-chain kernels, a debugging environment switch read once); re-entrant; the caller owns every buffer (the
 *     Python binding allocates the outputs exactly like the reference does with at::zeros);
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what
 *     the reference launches on);
 *   - return value: 0 on success, REGNET_ERR_* (<0) for an argument the reference would have
 *     rejected with TORCH_CHECK / CHECK_EQ / CHECK_GT / CHECK_GE, or a positive hipError_t from
 *     the launch (the reference's THCudaCheck(cudaGetLastError())).  Never throws.
 *   - float data is fp32, indices are int64, xyz tensors are the reference's channel-first
 *     (B,3,N) views addressed through explicit element strides (sb, sc, sn), so the
 *     non-contiguous views ScoreNet passes (score_network.py:46, pointnet2.py:89-90) need no copy.
 *
 * Reference interface each function replaces is cited as file:line relative to
 * /root/reference/multi_model/utils/pn2_utils/.
 */
#ifndef REGNET_HIP_H_
#define REGNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REGNET_OK 0
#define REGNET_ERR_SHAPE (-1)       /* CHECK_EQ / CHECK_GT / CHECK_GE style violation            */
#define REGNET_ERR_NULL (-2)        /* NULL pointer for a non-empty tensor                        */
#define REGNET_ERR_UNSUPPORTED (-3) /* size outside what this build supports (reported, not UB)  */

/* Library / build identification. */
int regnet_abi_version(void);
const char* regnet_build_info(void);
/* Human-readable text for a code returned by any entry point (static storage). */
const char* regnet_strerror(int code);

/* ---- pn2_ext.farthest_point_sample  (csrc/sampling.h:7-9, sampling_kernel.cu:126-170) --------
 * xyz (B,3,N) strided -> index (B,M) int64 contiguous.  index[b,0]=0; tie order is the
 * reference's (lane = j mod block, shared-memory tree).
 * Errors: M<=0 or N<M -> REGNET_ERR_SHAPE (CHECK_GT/CHECK_GE :136-137).                        */
int regnet_fps_f32(const float* xyz, int64_t sb, int64_t sc, int64_t sn, int64_t B, int64_t N,
                   int64_t M, int64_t* index, float* workspace, void* stream);
/* Scratch bytes regnet_fps_f32 needs for (B,N,M): 0 for N <= 4096 and for short runs (register-resident kernels);
 * B*N*4 for long runs (M >= 512 over 4096 < N <= 8192 points, 1024 <= M <= 8192 over 8192 < N <= 25600 points:
 * fps_cluster_kernel keeps the permutation of its in-kernel Morton sort there); beyond 25600 points B*N*4 (rounded
 * up to 16 bytes: the permutation, or the reference's `temp` tensor, sampling_kernel.cu:142) + 16640 bytes per scene
 * through which the 2-4 cooperating workgroups of a scene exchange their candidate records + 256 bytes (status word).  The callee
 * initialises it; `workspace` may be NULL when this returns 0.                                   */
int64_t regnet_fps_workspace_bytes(int64_t B, int64_t N, int64_t M);
/* regnet_fps_f32 for a cloud that is itself a furthest-point-sampling sequence -- what PointNet++ does at levels 2 and 3:
 * FarthestPointSampler over the previous level's centroids in their pick order (pointnet2.py:40-42, modules.py:23-26).
 * Pick k+1 of the producing run was the furthest point of the WHOLE cloud from picks 0..k, hence also the furthest among
 * the picks themselves, with the same fp32 distance: sampling the sequence from index 0 re-derives 0, 1, 2, ... unless two
 * points TIE at a maximum (the reference's tie order depends on array positions, which differ between the two runs).
 *   first_tie (B) int32, may be NULL: written with the first pick position (>= 1) of THIS run whose choice was not a unique
 *             strict maximum (several holders, or all distances zero); 0x7fffffff when there was none; 0 = "not tracked"
 *             (the single-workgroup round kernels for short runs and the streaming kernels; cooperating workgroups count
 *             every pick of their exact one-pick path).
 *   prefix_ok (B) int32, may be NULL: the `first_tie` of the run that PRODUCED this cloud's order.  Scenes with
 *             prefix_ok[b] >= M get index[b, :] = 0 .. M-1 without sampling (and first_tie[b] = prefix_ok[b]); the others are
 *             sampled as by regnet_fps_f32.  Results are identical to regnet_fps_f32's either way.
 * Same workspace and errors as regnet_fps_f32.                                                                           */
int regnet_fps_chain_f32(const float* xyz, int64_t sb, int64_t sc, int64_t sn, int64_t B, int64_t N, int64_t M,
                         int64_t* index, float* workspace, const int32_t* prefix_ok, int32_t* first_tie, void* stream);
/* Byte offset, inside that workspace, of the launch's 32-bit status word, or -1 when the kernel (B,N,M) selects has
 * none (every single-workgroup kernel).  The cooperative kernels (N > 25600) poll each other's exchange slots with a
 * bounded budget; a workgroup whose partner never answered ORs bit 0 into the word and stops sampling -- `index` is
 * then incomplete.  The callee zeroes the word; the caller reads it once the stream has passed the launch (the Python
 * binding accumulates it into a per-device flag and raises lazily: pn2_ext.raise_if_fps_failed).                      */
int64_t regnet_fps_status_offset_bytes(int64_t B, int64_t N, int64_t M);

/* ---- pn2_ext.ball_query  (csrc/ball_query.h:7-11, ball_query_kernel.cu:87-131) ---------------
 * xyz (B,3,N1) strided, centroids (B,3,N2) strided -> index (B,N2,K) int64, count (B,N2) int64.
 * First K points in index order with d2 < radius*radius (fp32, strict); first hit pre-fills all
 * K slots; a ball with no hit keeps index 0 / count 0 (the reference's at::zeros).              */
int regnet_ball_query_f32(const float* xyz, int64_t sb, int64_t sc, int64_t sn,
                          const float* centroids, int64_t cb, int64_t cc, int64_t cn, int64_t B,
                          int64_t N1, int64_t N2, float radius, int64_t K, int64_t* index,
                          int64_t* count, void* stream);

/* ---- uniform-grid variants of the two all-pairs scans (same results, fewer distance evaluations) ----
 * The source points (keys of the 3-NN search / the cloud of the ball query) are binned into cubic cells
 * inside a caller-provided workspace of regnet_grid_workspace_bytes(B, N_source) bytes (16-byte aligned);
 * outputs are bit-identical to regnet_three_nn_f32 / regnet_ball_query_f32.  The ball-query variant
 * supports K <= 64 (REGNET_ERR_UNSUPPORTED otherwise: use the plain entry point).                    */
int64_t regnet_grid_workspace_bytes(int64_t B, int64_t N_source);
int regnet_three_nn_grid_f32(const float* query, int64_t qb, int64_t qc, int64_t qn, const float* key,
                             int64_t kb, int64_t kc, int64_t kn, int64_t B, int64_t N1, int64_t N2,
                             int64_t* index, float* dist2, void* workspace, void* stream);
int regnet_ball_query_grid_f32(const float* xyz, int64_t sb, int64_t sc, int64_t sn,
                               const float* centroids, int64_t cb, int64_t cc, int64_t cn, int64_t B,
                               int64_t N1, int64_t N2, float radius, int64_t K, int64_t* index,
                               int64_t* count, void* workspace, void* stream);

/* ---- pn2_ext.group_points_forward / backward  (csrc/grouping.h:7-14) ------------------------
 * forward : input (B,C,N1) strided, index (B,N2,K) contiguous -> out (B,C,N2,K) contiguous.
 * backward: grad_out (B,C,N2,K) strided -> grad_in (B,C,N1) contiguous, zero-filled by callee. */
int regnet_group_points_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sn,
                                const int64_t* index, int64_t B, int64_t C, int64_t N1, int64_t N2,
                                int64_t K, float* out, void* stream);
int regnet_group_points_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn2,
                                int64_t sk, const int64_t* index, int64_t B, int64_t C, int64_t N1,
                                int64_t N2, int64_t K, float* grad_in, void* stream);

/* ---- pn2_ext.point_search (3-NN)  (csrc/interpolate.h:8-11, interpolate_kernel.cu:88-128) ----
 * query (B,3,N1), key (B,3,N2) strided -> index (B,N1,3) int64, dist2 (B,N1,3) fp32 ascending
 * SQUARED distances, earlier key wins ties.  N2<3 -> REGNET_ERR_SHAPE (CHECK_GE :102).          */
int regnet_three_nn_f32(const float* query, int64_t qb, int64_t qc, int64_t qn, const float* key,
                        int64_t kb, int64_t kc, int64_t kn, int64_t B, int64_t N1, int64_t N2,
                        int64_t* index, float* dist2, void* stream);

/* ---- pn2_ext.interpolate_forward / backward  (csrc/interpolate.h:13-22) ----------------------
 * forward : input (B,C,M) strided, index/weight (B,N,3) contiguous -> out (B,C,N) contiguous.
 * backward: grad_out (B,C,N) strided -> grad_in (B,C,M) contiguous, zero-filled by callee.      */
int regnet_interpolate_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sm,
                               const int64_t* index, const float* weight, int64_t B, int64_t C,
                               int64_t M, int64_t N, float* out, void* stream);
int regnet_interpolate_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn,
                               const int64_t* index, const float* weight, int64_t B, int64_t C,
                               int64_t M, int64_t N, float* grad_in, void* stream);

/* ---- dgcnn_ext.gather_knn_forward / backward  (functions/csrc/gather_knn.h) ------------------
 * Same gather / scatter-add as group_points with N2 == number of index rows.                    */
int regnet_gather_knn_fwd_f32(const float* input, int64_t sb, int64_t sc, int64_t sn,
                              const int64_t* index, int64_t B, int64_t C, int64_t N, int64_t NI,
                              int64_t K, float* out, void* stream);
int regnet_gather_knn_bwd_f32(const float* grad_out, int64_t sb, int64_t sc, int64_t sn2,
                              int64_t sk, const int64_t* index, int64_t B, int64_t C, int64_t N,
                              int64_t NI, int64_t K, float* grad_in, void* stream);

/* ---- region grouping (host Python loops in the reference) -----------------------------------
 * regnet_radius_group_f32: dataset_utils/get_regiondataset.py:279-295,:311-352.
 * pc (B,N,*) rows with xyz first (strides pb,pn in floats), centres (B,NC,*) likewise ->
 * cand (B,NC,cap) int32 = ascending indices of the points with d2 <= d2_threshold (inclusive),
 * count (B,NC) int32 = number of members (may exceed cap; only the first cap are stored).
 * The caller derives d2_threshold from the reference's `sqrt(d2) <= R` test (see
 * region_ops.sqrt_le_threshold).                                                                 */
int regnet_radius_group_f32(const float* pc, int64_t pb, int64_t pn, const float* centres,
                            int64_t cb, int64_t cn, int64_t B, int64_t N, int64_t NC,
                            float d2_threshold, int64_t cap, int32_t* cand, int32_t* count,
                            void* stream);

/* regnet_select_positive_f32: dataset_utils/get_regiondataset.py:354-434 (the `score > thr` mask +
 * torch.nonzero per scene of the centre picker).  pc (B,N,*) rows with xyz first (strides pb,pn),
 * score (B,N) (row stride sb) -> index (B,N) int64: first count[b] entries = ascending ids of the
 * points with score > threshold (strict); xyz_out (B,3,N) contiguous: their coordinates in that
 * order, slots >= count[b] repeat the first positive; count (B) int32.                          */
int regnet_select_positive_f32(const float* pc, int64_t pb, int64_t pn, const float* score, int64_t sb,
                               int64_t B, int64_t N, float threshold, int64_t* index, float* xyz_out,
                               int32_t* count, void* stream);

/* regnet_box_crop_f32: multi_model/gripper_region_network.py:508-544 (the per-grasp torch.nonzero
 * loop).  group_points (n,G,*) rows with xyz first (strides gb,gn), centre (n,3), rot (n,3,3)
 * row-major = [approach; axis_y; minor_normal], xlim/ylim (n) per-grasp half extents, zlim
 * scalar -> cand (n,G) int32 ascending in-box positions, count (n) int32.                       */
int regnet_box_crop_f32(const float* group_points, int64_t gb, int64_t gn, const float* centre,
                        const float* rot, const float* xlim, const float* ylim, float zlim,
                        int64_t n, int64_t G, int32_t* cand, int32_t* count, void* stream);

/* regnet_gather_max_f32: multi_model/gripper_region_network.py:388-395 + utils/pointnet2.py:167
 * (and :334-343 + pointnet2.py:232 for the refine stage): out[r,:] = max_g feat[rows[r,g],:].
 * feat (num_rows,F) row-major, rows (R,G) int64 -> out (R,F).                                    */
int regnet_gather_max_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows,
                          int64_t R, int64_t G, float* out, void* stream);
/* regnet_gather_max_scene_f32: the same with the reference's `index + b * N` (gripper_region_network.py:388, :334) formed in
 * the address: output row r pools the index list rows[rid], rid = row_ids ? row_ids[r] : r (int64 device, R entries), whose
 * entries are LOCAL point ids of scene rid / per_scene; global feature row = entry + (rid / per_scene) * scene_stride.
 * Negative entries are skipped.  Needs F % 4 == 0 and a 16-byte aligned feat (REGNET_ERR_UNSUPPORTED otherwise).          */
int regnet_gather_max_scene_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows, const int64_t* row_ids,
                                int64_t R, int64_t G, int64_t per_scene, int64_t scene_stride, float* out, void* stream);

/* ---- per-point shared-MLP contraction on fp32 MFMA (channels-last activations) ---------------
 * Replaces the cuDNN/cuBLAS 1x1-conv + BatchNorm + ReLU chains the reference runs for
 * SharedMLP (pn2_utils/nn/modules/mlp.py:55-114, conv.py:6-76) inside PointNetSAModule.forward
 * (pn2_utils/modules.py:210-246), PointnetFPModule.forward (:500-509) and the head
 * (utils/pointnet2.py:116-119).  W is packed [Npad][Kpad] (Npad multiple of 128, Kpad multiple of
 * 16, zero padded); scale/shift are the eval-mode BatchNorm folded to a per-channel affine.
 *
 * regnet_mlp_layer_f32: C[P,N] = act(scale * (A[P,Ka] . W^T) + shift); pool_group == 64 additionally
 * takes the max over every 64 consecutive rows (torch.max(x, 3), modules.py:245) -> C[P/64, N].  */
int regnet_mlp_layer_f32(const float* A, int64_t lda, int64_t Ka, const float* W, int64_t Kpad,
                         const float* scale, const float* shift, float* C, int64_t ldc, int64_t P,
                         int64_t N, int relu, int pool_group, void* stream);

/* regnet_mlp_layer_splitk_f32: regnet_mlp_layer_f32 (no pooling) for SKINNY problems -- few rows, long K, e.g. the
 * grasp-region heads on B*64 regions (pointnet2.py:165-197, :227-254), where a 128-row tile grid cannot fill the
 * chip.  The K range is cut into `ksplit` slices multiplied by different workgroups; the slices' raw partial sums go
 * to `workspace` (regnet_mlp_splitk_workspace_bytes(P, N, ksplit) bytes, 16-byte aligned) and a second kernel adds
 * them IN INDEX ORDER and applies the affine + ReLU, so the result is deterministic.  1 <= ksplit <= Kpad / 16.   */
int64_t regnet_mlp_splitk_workspace_bytes(int64_t P, int64_t N, int64_t ksplit);
int regnet_mlp_layer_splitk_f32(const float* A, int64_t lda, int64_t Ka, const float* W, int64_t Kpad,
                                const float* scale, const float* shift, float* C, int64_t ldc, int64_t P,
                                int64_t N, int relu, int64_t ksplit, void* workspace, void* stream);

/* regnet_sa_layer1_f32: first SharedMLP layer of a set-abstraction block with the grouping fused
 * into the operand load (no (B,C,M,K) tensor is materialised; reference: QueryGrouper.forward,
 * modules.py:39-56).  Row p = (b, m, k): A[p] = [feat[b, nbr[p], 0:Cf] | xyz[b,:,nbr[p]] -
 * xyz[b,:,ctr[b*M+m]]]; W columns must be packed in that order (features first).  feat element
 * (b,n,c) at feat[b*fb + n*fn + c*fc]; xyz (B,3,N) with strides (xb,xc,xn); nbr (B,M,group) int64,
 * ctr (B,M) int64.  -> C[B*M*group, N].                                                          */
int regnet_sa_layer1_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf,
                         const float* xyz, int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr,
                         const int64_t* ctr, int64_t B, int64_t M, int64_t group, const float* W,
                         int64_t Kpad, const float* scale, const float* shift, float* C, int64_t ldc,
                         int64_t N, int relu, void* stream);

/* regnet_sa_layer12_f32: like regnet_sa_layer1_f32 but with the FIRST TWO SharedMLP layers of the block in
 * one kernel, for blocks whose gathered input is narrow (Cf + 3 <= 8, i.e. the level-1 SA of ScoreNet:
 * 3 rgb + 3 xyz).  Layer 1 (W1 packed [C1][8] with columns [feat | xyz | 0], folded BN scale1/shift1,
 * ReLU) runs on the VALU inside the operand load of layer 2 (W packed [Npad][Kpad = C1]); the (P x C1)
 * layer-1 activation never reaches HBM.  pool_group as in regnet_mlp_layer_f32.                      */
int regnet_sa_layer12_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf,
                          const float* xyz, int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr,
                          const int64_t* ctr, int64_t B, int64_t M, int64_t group, const float* W1,
                          const float* scale1, const float* shift1, int64_t C1, const float* W,
                          int64_t Kpad, const float* scale, const float* shift, float* C, int64_t ldc,
                          int64_t N, int relu, int pool_group, void* stream);

/* regnet_sa_chain3_f32: a WHOLE narrow-input set-abstraction block in one kernel (pn2_utils/modules.py:210-246
 * for the level-1 block of PointNet2Seg, pointnet2.py:40-42): gather [feat | xyz - centre] (Cf + 3 <= 8) ->
 * layer 1 (W1 [C1][8], VALU) -> layer 2 (W2 packed [128][K2pad = 128]) -> layer 3 (W3 packed [C3pad][K3pad = 128],
 * C3 % 32 == 0) -> max over the group's 64 neighbours -> out (B*M, ldo).  Layers 1 and 2 always apply their folded
 * BN affine + ReLU, layer 3 its affine and ReLU if relu3.  The products are formed transposed (channels x points)
 * so each layer's MFMA accumulator is the next layer's B operand: no activation is written to LDS or HBM.
 * `count` (B*M int64, optional): the ball query's member counts -- slots >= count of a neighbourhood repeat slot 0
 * (regnet_ball_query_f32), so a neighbourhood with <= 32 members needs only its first 32 slots and half the MFMAs,
 * and two neighbourhoods with 33..48 members that sit 4 apart in a workgroup's 8 (slots 8 g + w and 8 g + w + 4 of
 * the processing order) share their third 32-row tile: three tiles of MFMAs for the pair instead of four.
 * `order` (B*M int64, optional): a permutation of the neighbourhoods giving the order in which workgroups (8
 * neighbourhoods each) take them; sorting by member-count class (<= 32, 33..48, more) makes whole workgroups of
 * one kind.  Outputs do not depend on `count` / `order` (bit-identical with and without).
 * Supported: group == 64, C1 == C2 == 128; anything else returns REGNET_ERR_UNSUPPORTED (use the layer-wise
 * entry points).  Same values as regnet_sa_layer12_f32 + regnet_mlp_layer_f32(pool) up to fp32 summation order. */
int regnet_sa_chain3_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf, const float* xyz,
                         int64_t xb, int64_t xc, int64_t xn, const int64_t* nbr, const int64_t* ctr,
                         const int64_t* count, const int64_t* order, int64_t B, int64_t M, int64_t group,
                         const float* W1, const float* scale1,
                         const float* shift1, int64_t C1, const float* W2, int64_t K2pad,
                         const float* scale2, const float* shift2, int64_t C2, const float* W3,
                         int64_t K3pad, const float* scale3, const float* shift3, int64_t C3, int relu3,
                         float* out, int64_t ldo, void* stream);

/* regnet_pack_rows_f32: channels-last rows [feature(Cf) | xyz(3) | 0...] (width W >= Cf + 3) of all B*N source points,
 * the operand of the per-source-point first layer (U, V of regnet_sa_premul_layer_f32); feat (B,Cf,N) and xyz (B,3,N)
 * with element strides (b, c, n).  Replaces a zero fill and two strided copies.                                    */
int regnet_pack_rows_f32(const float* feat, int64_t fb, int64_t fc, int64_t fn, int64_t Cf, const float* xyz,
                         int64_t xb, int64_t xc, int64_t xn, int64_t B, int64_t N, int64_t W, float* out,
                         void* stream);
/* regnet_pack_rows_centred_f32: the same with the xyz columns written as xyz - mu[b] (mu (B,3) contiguous device floats, or
 * NULL = regnet_pack_rows_f32): the scene-mean centring of the pre-multiplied first layer (fused.PREMUL_CENTRE) inside the
 * pack instead of as a tensor subtraction in front of it -- one fp32 subtraction per value either way, same bits.          */
int regnet_pack_rows_centred_f32(const float* feat, int64_t fb, int64_t fc, int64_t fn, int64_t Cf, const float* xyz,
                                 int64_t xb, int64_t xc, int64_t xn, const float* mu, int64_t B, int64_t N, int64_t W,
                                 float* out, void* stream);

/* ---- gather_points (multi_model/utils/pn2_utils/function.py:11-26; callers modules.py:41, :238) ---------------------
 * out[b][c][m] = points[b][c][index[b][m]]: points element (b,c,n) at points[b*pb + c*pc + n*pn], index (b,m) at
 * index[b*ib + m*im] (int64), out element (b,c,m) at out[b*ob + c*oc + m*om].  An index outside [0, N) writes 0 and ORs 1
 * into *status (device int32, may be NULL) -- torch.gather raises for it; the Python wrapper checks lazily.                  */
int regnet_gather_points_f32(const float* points, int64_t pb, int64_t pc, int64_t pn, int64_t B, int64_t C, int64_t N,
                             const int64_t* index, int64_t ib, int64_t im, int64_t M, float* out, int64_t ob, int64_t oc,
                             int64_t om, int32_t* status, void* stream);

/* regnet_class_order_i64: order (n) int64 = the stable sort permutation of the level-1 neighbourhoods by cost class
 * (count > 32) + (count > 48) -- what torch.argsort(class, stable=True) returns (fused.chain3_order; the processing order
 * of regnet_sa_chain3_f32) -- as a three-bin counting sort in one launch.  count (n) int64 device (regnet_ball_query_f32's
 * second output), n <= 2^24.                                                                                                 */
int regnet_class_order_i64(const int64_t* count, int64_t n, int64_t* order, void* stream);

/* regnet_grasp_collision_counts_f32 / regnet_grasp_antipodal_stats_f32: the per-grasp point scans of the reference's grasp
 * evaluation, dataset_utils/eval_score/eval.py:4-24 -> eval_utils/evaluation_data_generator.py (EvalDataTest /
 * EvalDataValidate .finger_hand_view :188-236 / :420-483, .finger_hand_scene :485-537, ._antipodal_score :392-418;
 * constants: eval_score/configs/config.py), which loop over the grasps in Python.  test.py:147 applies the view-cloud
 * filter to every predicted grasp (utils.py:391-401); utils.py:270-295 scores validation grasps against the scene cloud.
 * points (and normals): N rows, element strides (pn, pc) / (nn, nc) for (point, coordinate); T (B,4,4) row-major
 * global->local matrices (:91-93).  With (x,y,z) = T[b] (p,1) and all comparisons strict, counts[b] (4 x int32) =
 *   { #(x_lo < x < x_hi), # behind the hand (.. and |y| < half_width, x < back_x, |z| < half_thickness),
 *     # inside a finger (.. |z| < half_thickness, half_space < |y| < half_width), # closing region (.. |y| < half_space) };
 * x_hi_per_grasp (B) overrides x_hi when not NULL (per-grasp depths, :428-430).  The caller applies the reference's
 * thresholds.  Antipodal statistics over the closing-region points: stats[b] = { y_max, y_min, sum |n_y| over
 * y > y_max - d, sum |n_y| over y < y_min + d } with d = min((y_max - y_min) / 3, neighbour_depth) and n_y the normal's
 * y component in the grasp frame; side_counts[b] = the two point counts (the reference's score is the product of the
 * two means).  fp32, individually rounded, no contraction; the sums are accumulated in fp64.                            */
int regnet_grasp_collision_counts_f32(const float* points, int64_t pn, int64_t pc, int64_t N, const float* T, int64_t B,
                                      float x_lo, float x_hi, const float* x_hi_per_grasp, float half_thickness,
                                      float half_width, float half_space, float back_x, int32_t* counts, void* stream);
int regnet_grasp_antipodal_stats_f32(const float* points, int64_t pn, int64_t pc, const float* normals, int64_t nn,
                                     int64_t nc, int64_t N, const float* T, int64_t B, float x_lo, float x_hi,
                                     const float* x_hi_per_grasp, float half_thickness, float half_width, float half_space,
                                     float back_x, float neighbour_depth, float* stats, int32_t* side_counts, void* stream);

/* Training twins (round 3).  regnet_gather_max_arg_f32: regnet_gather_max_f32 that also returns, per (row, channel), the
 * source row that gave the maximum (arg (R,F) int64; first maximum in group order; a negative row id counts from the end as
 * the reference's `all_feature.view(-1,F)[index]` does, gripper_region_network.py:382-390) -- the backward of the pooled
 * region feature then scatters R x F values.  regnet_rowsum_neg_f32: out[r] = -sum_k x[r*K + k] (K a power of two in
 * 4..256): the gradient of the per-centre term of a pre-multiplied first layer (dV = -sum over the K neighbours of dY). */
int regnet_gather_max_arg_f32(const float* feat, int64_t num_rows, int64_t F, const int64_t* rows, int64_t R, int64_t G,
                              float* out, int64_t* arg, void* stream);
/* regnet_scatter_max_grad_f32: the backward of regnet_gather_max_arg_f32 -- grad[arg[r][f]][f] += dy[r][f] for dy, arg (R, F)
 * (arg < 0: nothing), by float atomics.  Row `row` is scene b = row / scene_rows, point n = row % scene_rows and lands at
 * grad + b * batch_stride + n * row_stride + f * ch_stride: (scene_rows = total rows, row_stride F, ch_stride 1) for a
 * (rows, F) matrix, (scene_rows N, batch_stride F * N, row_stride 1, ch_stride N) for the channel-first (B, F, N) gradient
 * of the feature map itself.  grad is ADDED to (zero it, or hand in a gradient that is to receive this one).           */
int regnet_scatter_max_grad_f32(const float* dy, const int64_t* arg, int64_t R, int64_t F, int64_t scene_rows,
                                int64_t batch_stride, int64_t row_stride, int64_t ch_stride, float* grad, void* stream);
int regnet_rowsum_neg_f32(const float* x, int64_t rows, int64_t K, float* out, void* stream);

/* regnet_heads_chain_f32: a small tree of conv(1x1) + folded eval BatchNorm (+ ReLU) layers over FEW rows in one launch -- the
 * grasp heads of the region stage (pointnet2.py:174-188: 256 -> 1024 -> {256 -> 128 -> k_cls | 256 -> 128 -> k_reg};
 * pointnet2.py:240-253: 384 -> 1024 -> {128 -> k_cls | 128 -> k_reg}) on the B * 64 centres / the valid crops.  A workgroup
 * takes 16 rows through all layers; activations stay in LDS.  x (n, Kx) rows with stride ldx (multiple of 4, 16-byte
 * aligned); descr: `layers` (<= 8) records of 9 int64 [W, scale, shift (device addresses: W packed [>= ceil16(N)][Kpad] zero
 * padded, scale / shift [N]: y = relu?(scale * (W . x) + shift)), K, Kpad (multiple of 16), N, relu, src, dst]; buffers: 0 =
 * input, 1..3 = LDS scratch (a scratch buffer's width is its producer's N rounded up to 16 and must equal its consumers'
 * Kpad), dst 4 / 5 = out_a (n, lda) / out_b (n, ldb).  Layers run in order.  REGNET_ERR_UNSUPPORTED beyond 160 KiB of LDS. */
int regnet_heads_chain_f32(const float* x, int64_t ldx, int64_t Kx, int64_t n, const int64_t* descr, int64_t layers,
                           float* out_a, int64_t lda, float* out_b, int64_t ldb, void* stream);

/* regnet_heads_tree_f32: the same two trees for MANY rows (the 512 centres of 8 scenes, the 4000 of test.py:68) in one launch.
 * A workgroup takes 32 rows; the wide trunk activation (pointnet2.py:174 / :240: 1024 columns) is produced 256 columns at a
 * time and consumed at once by BOTH branches' first layers, whose weights the caller concatenates row-wise (N2 = 512 for
 * 1024 -> 256 | 256, 256 for 1024 -> 128 | 128); their outputs accumulate in registers, so no row count needs the trunk in
 * LDS and every weight fragment feeds two row blocks.  Same operand mapping and K order as regnet_heads_chain_f32: same bits.
 * descr: (2 + tails) records of 11 int64 [W, scale, shift (device addresses, packed as above), K, Kpad, N, relu, src,
 * src_off, dst, dst_off]; record 0 = trunk (K = Kx, Kpad = ceil16(Kx), N = Nt a multiple of 256), record 1 = the joined
 * first layers (K = Kpad = Nt, N = N2 in {256, 512}), then tails <= 4 in execution order with src 0 = the stage-2 rows / 1 =
 * the tail buffer read from column src_off (multiple of 4) on, dst 1 = the tail buffer at column dst_off, 4 / 5 = out_a (n,
 * lda) / out_b (n, ldb).  REGNET_ERR_UNSUPPORTED beyond 160 KiB of LDS.                                                  */
int regnet_heads_tree_f32(const float* x, int64_t ldx, int64_t Kx, int64_t n, const int64_t* descr, int64_t tails,
                          float* out_a, int64_t lda, float* out_b, int64_t ldb, void* stream);

/* The same heads in TRAINING mode, one layer per call (pointnet2.py:174-188, :240-253: nn.Conv1d(K, N, 1) with bias ->
 * nn.BatchNorm1d(N) on batch statistics -> [ReLU], on the R labelled centres / valid crops of the iteration): six launches
 * per layer forward and seven backward through torch, on a stream the host paces.
 * regnet_head_layer_train_supported: 2 <= R <= 1024, K a multiple of 64, N >= 1.
 * regnet_head_layer_train_fwd_f32: X (R, K) rows with stride ldx (multiple of 4; X and W 16-byte aligned), W (N, K), bias
 *   (N) or NULL -> xhat (R, N) the normalised pre-affine activation, Y (R, N) = [relu](gamma * xhat + beta), save_invstd (N);
 *   running_mean / running_var (both or neither) are updated in place with `momentum` (running_var from the unbiased
 *   variance, running_mean from mean(X . W^T) + bias), *num_batches_tracked (device int64, may be NULL) is incremented.
 * regnet_head_layer_train_bwd_f32: dY (R, N) with row stride lddy = the gradient of Y; Y is read for the ReLU mask only
 *   (may be NULL when relu == 0) -> dZ (R, N) the gradient of the convolution's output (workspace of the call), dW (N, K),
 *   dbias (N, may be NULL), dgamma (N), dbeta (N), and, when dX != NULL, dX (R, K) with row stride lddx = dZ . W, ADDED to
 *   its contents when accumulate_dx != 0 (the second branch of a fork).  Deterministic: fixed summation orders. */
int regnet_head_layer_train_supported(int64_t R, int64_t K, int64_t N);
int regnet_head_layer_train_fwd_f32(const float* X, int64_t ldx, const float* W, const float* bias, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    float momentum, float eps, int64_t R, int64_t K, int64_t N, int relu, float* xhat, float* Y,
                                    float* save_invstd, void* stream);
int regnet_head_layer_train_bwd_f32(const float* dY, int64_t lddy, const float* Y, const float* xhat, const float* gamma,
                                    const float* save_invstd, const float* X, int64_t ldx, const float* W, int64_t R, int64_t K,
                                    int64_t N, int relu, float* dZ, float* dW, float* dbias, float* dgamma, float* dbeta,
                                    float* dX, int64_t lddx, int accumulate_dx, void* stream);

/* Decodes of the two grasp heads WITHOUT labels (inference), one launch each instead of ~25 / ~12 small tensor ops.
 * regnet_stage2_decode_f32 (gripper_region_network.py:69-90, `ground is None`): cls (n,A), reg (n,A,C >= 7), centre rows
 * (n, centre_ld >= 3), tmpl (A,4) = the anchors' [r | theta] -> out (n,C): the arg-max anchor's regression turned into
 * [delta*radius + centre | (delta + r)/sqrt(|.|^2 + 1e-12) | pi*(delta + theta) | channels 7..], the latter through a
 * sigmoid when sigmoid_tail != 0 (pointnet2.py:187, for a head that hands over raw values).
 * regnet_refine_decode_f32 (gripper_region_network.py:201-215): grasp (n, grasp_ld >= C), cls (n,2), reg (n,C >= 8) ->
 * final_grasp (n,C) = grasp + deltas (the first three times radius), flags (2,n) uint8: [class 1 | class 1 and
 * final[:,7] > score_thre].                                                                                            */
int regnet_stage2_decode_f32(const float* cls, const float* reg, int64_t A, int64_t C, const float* centre,
                             int64_t centre_ld, const float* tmpl, float radius, int sigmoid_tail, int64_t n, float* out,
                             void* stream);
int regnet_refine_decode_f32(const float* grasp, int64_t grasp_ld, const float* cls, const float* reg, int64_t C,
                             float radius, float score_thre, int64_t n, float* final_grasp, uint8_t* flags, void* stream);

/* The two grasp losses of the training iteration with their gradients (csrc/losses.hip; gripper_region_network.py:46-184,
 * :186-309 with labels).  Every row-wise quantity in one launch, the class-balanced cross entropy in a second one after the
 * host's numpy draws; smooth-L1 is torch's default (beta 1), 1 - cos as compute_cos_sim / CosineEmbeddingLoss write it.
 * regnet_stage2_loss_rows_f32: cls (n,A), reg (n,A,10), centre rows (n, centre_ld), tmpl (A,4), label rows (n, label_ld >= 10),
 *   rows (m) int64 = the labelled centres (NULL: all n = m), weights4 = the four regression terms' weights (host pointer) ->
 *   compact per labelled centre: next_grasp (m,10) (arg-max decode), pick / g8 (m) int32 (arg-max class / label's anchor),
 *   a_gt (m,7), terms (m,12) = [4 regression terms | 4 monitoring terms | g8 == pick | 0 0 0]; dreg (n,A,10) rows `rows` =
 *   gradient of the weighted regression terms (other rows untouched: zero-fill first).
 * regnet_ce_rows_f32: loss[k] = CE(cls[rows[idx[k]]], target[idx[k]]), dcls[rows[idx[k]], :] = (softmax - onehot) * scale.
 * regnet_refine_loss_rows_f32: grasp rows (m, ld), cls (m,2), reg (m,10), label rows -> final_grasp (m,10), flags (3,m) uint8
 *   [class 1 | class 1 and score kept | label-positive], terms (m,20) = [4 regression terms of positive rows | 3 x 4
 *   monitoring terms | 4 confusion counts], dreg (m,10) = SL1' of the positive rows' residuals (unscaled).            */
/* regnet_label_match_f32: the training labels of the centres (dataset_utils/get_regiondataset.py:45-134, :136-199 with
 * use_theta): packed (B, Gmax, 19) = the scenes' ground-truth grasps padded to the longest record [16 row-major entries of the
 * 4x4 frame | score | antipodal score | centre score], gcount (B) their counts, centre rows at centre + b * centre_sb + c *
 * centre_sn (xyz first) -> out (B, Nc, 10) = [grasp centre | closing axis with x >= 0 | angle in (-pi, pi] | 3 scores] of the
 * grasp whose contact point is nearest (the reference's fp32 expansion compared as float64, first minimum), or the reference's
 * filler (-1, axis +1) when that squared distance exceeds max_sq; wide_row (B * Nc) = 1 where the row's antipodal score is not
 * -1 (the reference returns 8 channels when none is).  Replaces ~60 tensor launches. */
int regnet_label_match_f32(const float* packed, const int32_t* gcount, int64_t Gmax, const float* centre, int64_t centre_sb,
                           int64_t centre_sn, int64_t B, int64_t Nc, float depth, double max_sq, float* out,
                           int32_t* wide_row, void* stream);
int regnet_stage2_loss_rows_f32(const float* cls, const float* reg, int64_t A, int64_t C, const float* centre,
                                int64_t centre_ld, const float* tmpl, const float* label, int64_t label_ld, float radius,
                                const float* weights4, const int64_t* rows, int64_t m, float* next_grasp, int32_t* pick,
                                int32_t* g8, float* a_gt, float* terms, float* dreg, void* stream);
int regnet_ce_rows_f32(const float* cls, int64_t A, const int32_t* target, const int64_t* idx, const int64_t* rows,
                       int64_t nb, float scale, float* loss, float* dcls, void* stream);
int regnet_refine_loss_rows_f32(const float* grasp, int64_t grasp_ld, const float* cls, const float* reg, const float* label,
                                int64_t label_ld, int64_t C, float radius, float score_thre, int64_t m, float* final_grasp,
                                uint8_t* flags, float* terms, float* dreg, void* stream);

/* regnet_gripper_frame_f32: grasp (n, ld >= 7) rows [centre | closing axis | theta | ...] -> centre (n,3), rot (n,3,3) with rows
 * [approach; axis_y; minor_normal] -- the frame maths of get_gripper_region_transform (gripper_region_network.py:447-506) in
 * one launch.  regnet_crop_pick: the drawn candidate positions of a box crop resolved to group positions and scene indices
 * (index / index_inall (n,R) int64; -1 rows for grasps whose crop is invalid; :540-548). */
int regnet_gripper_frame_f32(const float* grasp, int64_t ld, int64_t n, float* centre, float* rot, void* stream);
int regnet_crop_pick(const int32_t* cand, int64_t G, const int64_t* pos, int64_t R, const uint8_t* valid,
                     const int64_t* group_index, int64_t gi_stride, int64_t n, int64_t* index, int64_t* index_inall,
                     void* stream);

/* regnet_resample_groups_f32: the gather half of get_regiondataset.py:331-352 (_get_group_pc).  cand (B,Nc,cap) int32: the
 * ascending member lists of regnet_radius_group_f32; pos (B,Nc,G) int64: positions into them drawn on the host with numpy's
 * RNG stream (-1 in every slot of a centre without candidates); pc (B,N,C) rows with element strides (pb, pn), channels
 * contiguous.  index[b,c,g] = cand[b,c,pos[b,c,g]] and points[b,c,g,:] = pc[b,index,:], or -1 / -1.0f where pos < 0 (the
 * reference fills empty groups with -1, :346-350).  Replaces seven tensor ops per pass.
 * A position >= cap or a candidate outside [0, N) (an upstream count / capacity mismatch; the torch.gather this replaces
 * raised on it) reads nothing, yields -1 and ORs 1 into *out_of_range (device int32, may be NULL).                       */
int regnet_resample_groups_f32(const float* pc, int64_t pb, int64_t pn, int64_t C, const int32_t* cand, int64_t cap,
                               const int64_t* pos, int64_t B, int64_t Nc, int64_t G, int64_t N, int32_t* out_of_range,
                               int64_t* index, float* points, void* stream);

/* regnet_estimate_normals_f32: surface normals of a scene cloud for validation records that carry no scene_normal --
 * dataset_utils/eval_score/eval_utils/torch_scene_point_cloud.py:17-19 -> eval_utils/pointcloud.py:27-43, i.e. open3d's
 * estimate_normals(KDTreeSearchParamHybrid(radius = NORMAL_RADIUS, max_nn = NORMAL_MAX_NN)), normalize_normals and
 * orient_normals_towards_camera_location (configs/config.py:16-17).  open3d is a dependency outside the reference tree
 * (absent from this image: parity unpinned, see oracle/collision_oracle.py); the algorithm restated is its published one:
 * neighbours of point i = the max_nn nearest points (itself included) with squared distance -- double,
 * ((dx*dx)+(dy*dy))+(dz*dz) -- strictly below float(radius*radius), ties ranked by index; fewer than 3 neighbours ->
 * (0,0,1); otherwise the eigenvector of the smallest eigenvalue of the neighbours' covariance (second moments / n -
 * mean products, double), a zero vector replaced by (0,0,1); normalised; negated when it points away from the camera
 * (normal . (camera - point) < 0).  xyz (N,3) contiguous float32, normals (N,3) float32, count (N) int32 = neighbours
 * used (may be NULL).  workspace: regnet_normals_workspace_bytes(N) bytes, 16-byte aligned.  max_nn <= 64.             */
int64_t regnet_normals_workspace_bytes(int64_t N);
int regnet_estimate_normals_f32(const float* xyz, int64_t N, double radius, int64_t max_nn, double cam_x, double cam_y,
                                double cam_z, float* normals, int32_t* count, void* workspace, void* stream);

/* regnet_bn_relu_train_fwd_f32 / _bwd_f32: TRAINING-mode BatchNorm (+ ReLU, + max over the K neighbours) of a shared-MLP
 * block -- nn/modules/conv.py:30-36, :70-76 (bn then relu after the bias-free 1x1 convolution) and the set-abstraction
 * reduction torch.max(new_feature, 3) of modules.py:245 -- as two HBM passes each way instead of torch's 13-18.
 * x (B, C, L) contiguous, 16-byte aligned; statistics per channel over (B, L), biased variance for the normalisation,
 * running_mean / running_var (may be NULL) updated with `momentum` and the unbiased variance, save_mean / save_invstd
 * (C) written for the backward.  pool_group == 0: y (B, C, L) = [relu](bn(x)).  pool_group = K (a power of two in
 * 4..256 dividing L, the K neighbours innermost): y (B, C, L/K) = max over each run of K, pool_index (B, C, L/K) int32 =
 * position of the selected element (smallest on ties).  workspace: regnet_bn_workspace_bytes(C) bytes.
 * Backward: dy shaped like y; dx (B, C, L), dgamma / dbeta (C) are overwritten; the ReLU mask is recomputed from x
 * (pooled: taken from y, which must then be the forward's output).                                                   */
int64_t regnet_bn_workspace_bytes(int64_t C);
int regnet_bn_relu_train_fwd_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, int relu,
                                 int64_t pool_group, float* y, int32_t* pool_index, float* save_mean, float* save_invstd,
                                 void* workspace, void* stream);
int regnet_bn_relu_train_bwd_f32(const float* x, const float* y, const float* dy, const int32_t* pool_index, int64_t B,
                                 int64_t C, int64_t L, const float* gamma, const float* beta, const float* save_mean,
                                 const float* save_invstd, int relu, int64_t pool_group, float* dx, float* dgamma,
                                 float* dbeta, void* workspace, void* stream);

/* The statistics half of regnet_bn_relu_train_fwd_f32 alone, plus the normalisation as a per-channel affine
 * (scale = gamma * invstd, shift = beta - mean * scale), for the convolution that consumes the BatchNorm's output and applies
 * it to its own operand (regnet_conv1x1_fwd_bnrelu_stream_f32 / regnet_conv1x1_wgrad_bnrelu_f32 below): conv -> bn -> relu ->
 * conv of nn/modules/mlp.py:95-107 without writing the normalised activation.  The backward is regnet_bn_relu_train_bwd_f32
 * on the input gradient of that convolution.                                                                           */
int regnet_bn_train_stats_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                              float* scale, float* shift, void* workspace, void* stream);
/* ..._from_sums: the same two entry points with the statistics PASS already done -- `workspace` holds, per channel, the sum and
 * the sum of squares of x over all B L elements (2 C doubles) as regnet_conv1x1_fwd_stats_stream_f32, the convolution that
 * produced x, accumulated them from its output tiles: x is not read for its statistics (regnet_bn_train_stats_from_sums_f32
 * does not take it at all).                                                                                            */
int regnet_bn_relu_train_fwd_from_sums_f32(const float* x, int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta,
                                           float eps, float momentum, float* running_mean, float* running_var, int relu,
                                           int64_t pool_group, float* y, int32_t* pool_index, float* save_mean,
                                           float* save_invstd, void* workspace, void* stream);
int regnet_bn_train_stats_from_sums_f32(int64_t B, int64_t C, int64_t L, const float* gamma, const float* beta, float eps,
                                        float momentum, float* running_mean, float* running_var, float* save_mean,
                                        float* save_invstd, float* scale, float* shift, void* workspace, void* stream);

/* ---- set-abstraction layers 1+2 with layer 1 evaluated per SOURCE point ---------------------------
 * The first SharedMLP layer of a set-abstraction block (pn2_utils/modules.py:44-55: conv over
 * [xyz_j - xyz_c | feature_j]) is linear in the gathered row, so
 *   scale1 * W1 [f_j | x_j - x_c] + shift1 = U[j] - V[c],   U = scale1 * W1 [f | x]  (B*Nsrc rows),
 *                                                             V = scale1 * W1x x_c - shift1 (B*M rows),
 * both produced by regnet_mlp_layer_f32 (relu = 0).  This entry point gathers, applies the ReLU and
 * multiplies by layer 2:  A[p][k] = max(U[b*Nsrc + nbr[p]][k] - V[p / group][k], 0), k < C1 (C1 % 4 == 0);
 * C = epilogue(A . W^T) exactly as regnet_mlp_layer_f32 (pool_group 0 or 64).  Same values as the
 * gather-then-multiply form up to fp32 rounding of the re-associated sum.                          */
int regnet_sa_premul_layer_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, int64_t C1,
                               const int64_t* nbr, int64_t B, int64_t Nsrc, int64_t M, int64_t group,
                               const float* W, int64_t Kpad, const float* scale, const float* shift,
                               float* C, int64_t ldc, int64_t N, int relu, int pool_group, void* stream);

/* regnet_interp_concat_f32: FeatureInterpolator.forward (modules.py:104-131) channels-last:
 * out[b*Nd+n] = [sum_k w_k * sparse[b, idx[b,n,k], 0:Cs] | dense[b,n,0:Cd] | 0...], w from squared
 * distances (inv = 1/max(d2,eps), w = inv/sum).  sparse (b,n,:) at sparse[b*sb + n*sn + c];
 * dense element (b,n,c) at dense[b*db + n*dn + c*dc]; out row stride ldo >= Cout >= Cs+Cd.        */
int regnet_interp_concat_f32(const float* sparse, int64_t sb, int64_t sn, int64_t Cs,
                             const int64_t* idx, const float* dist2, float eps, const float* dense,
                             int64_t db, int64_t dn, int64_t dc, int64_t Cd, int64_t B, int64_t Nd,
                             float* out, int64_t ldo, int64_t Cout, void* stream);

/* ---- feature propagation with the first SharedMLP layer applied BEFORE the interpolation -------------
 * The layer is linear in [sum_k w_k sparse[idx_k] | dense] (pn2_utils/modules.py:117-131), hence
 *   out[p][n] = relu(scale[n] * (sum_k w_k Ys[b, idx_k[p]][n] + Yd[p][n]) + shift[n]),
 * Ys (B,Ns,C) = Ws . sparse per SPARSE point (strides sb,sn; channel stride 1), Yd (B*Nd, ldd) = Wd . dense
 * per dense point or NULL, both from regnet_mlp_layer_f32 with scale 1 / shift 0 / relu 0; w_k as in
 * regnet_interp_concat_f32 (from squared distances).  A narrow skip input (Cd_small <= 4 channels, e.g.
 * rgb; element (b,c,n) at dense_small[b*db + c*dc + n*dn]) is multiplied in place with its weight columns
 * Wd4 (C x 4, zero padded) instead of going through Yd; pass NULL when unused.
 * C % 4 == 0, C/4 must divide 256.  out (B*Nd, ldo).                                                   */
int regnet_interp_affine_f32(const float* ys, int64_t sb, int64_t sn, const int64_t* idx,
                             const float* dist2, float eps, const float* yd, int64_t ldd,
                             const float* dense_small, int64_t db, int64_t dn, int64_t dc,
                             int64_t Cd_small, const float* Wd4, const float* scale, const float* shift,
                             int relu, int64_t B, int64_t Nd, int64_t C, float* out, int64_t ldo,
                             void* stream);

/* regnet_score_head_f32: score = sigmoid(bn_score(conv_score(x))) (utils/pointnet2.py:117-119),
 * x (P,C) channels-last, w (C), bn folded to (bn_scale, bn_shift).                                */
int regnet_score_head_f32(const float* x, int64_t ldx, int64_t C, const float* w, float bias,
                          float bn_scale, float bn_shift, float* score, int64_t P, void* stream);

/* regnet_sa_premul_chain_f32: layers 2 and 3 + the max over the 64 neighbours of the level-2 set-abstraction block of
 * PointNet2Seg (utils/pointnet2.py:40-42: sa_channels[1] = (256, 256, 512); pn2_utils/modules.py:39-56, :244-245) in one
 * kernel, on pre-multiplied layer-1 rows as regnet_sa_premul_layer_f32 takes them:
 *     x0[p] = relu(U[b * Nsrc + nbr[p]] - V[p / 64])  ->  256 -> 256 -> 512 (folded BN affine, ReLU; ReLU of the last
 *     layer if relu3)  ->  out[p / 64] = max over the neighbourhood.
 * U (B*Nsrc, ldu >= 256), V (B*M, ldv >= 256), nbr (B, M, 64) int64, out (B*M, ldo >= 512).  Activations stay in
 * registers (csrc/rowchain.hip); the (B*M*64 x 256) layer-2 activation of the two-launch path is never written.
 * `stream_w`: regnet_sa_premul_chain_stream_floats() floats = 24 stages [32 output channels][256 k] -- the 8 row
 * blocks of W2, then the 16 of W3 -- chunk-swizzled like regnet_fp_head_chain_f32's; `affine` = [scale2 | shift2 |
 * scale3 | shift3] (1536 floats); `ticket`: one zeroed int32 (work-queue head).  Same values as
 * regnet_sa_premul_layer_f32 + regnet_mlp_layer_f32(pool) up to fp32 summation order.                              */
int64_t regnet_sa_premul_chain_stream_floats(void);
int regnet_sa_premul_chain_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, const int64_t* nbr, int64_t B,
                               int64_t Nsrc, int64_t M, const float* stream_w, int64_t n_stages, const float* affine,
                               int64_t affine_floats, int relu3, float* out, int64_t ldo, int32_t* ticket,
                               void* stream);

/* regnet_sa3_premul_chain_f32: the same for the level-3 block (utils/pointnet2.py:40-42: 515 -> 512 -> 512 -> 1024 over
 * 256 x 64 rows per scene): U (B*Nsrc, >=512), V (B*M, >=512) pre-multiplied layer-1 rows as for
 * regnet_sa_premul_layer_f32; layer 2 (512 -> 512) runs as two K-halves so that its 512-wide activation stays in
 * registers, layer 3 (512 -> 1024) point-major with the max over the 64 neighbours; out (B*M, 1024).
 * `stream_w`: regnet_sa3_premul_chain_stream_floats() floats = 96 stages [32 output channels][256 k]: W2 as
 * (K-half, 16 row blocks), then W3 as (32 row blocks, K-half); `affine` = [scale2 | shift2 | scale3 | shift3]
 * (3072 floats).  Same values as regnet_sa_premul_layer_f32 + regnet_mlp_layer_f32(pool) up to fp32 summation order. */
int64_t regnet_sa3_premul_chain_stream_floats(void);
int regnet_sa3_premul_chain_f32(const float* U, int64_t ldu, const float* V, int64_t ldv, const int64_t* nbr, int64_t B,
                                int64_t Nsrc, int64_t M, const float* stream_w, int64_t n_stages, const float* affine,
                                int64_t affine_floats, int relu3, float* out, int64_t ldo, int32_t* ticket,
                                void* stream);

/* regnet_fp_head_chain_f32: the tail of the last feature-propagation block and the whole segmentation head of
 * PointNet2Seg as ONE kernel (utils/pointnet2.py:64-84 with fp_channels[2] = (256, 256, 256), :116-119 with
 * seg_channels = (512, 256, 256, 128); pn2_utils/modules.py:500-509):
 *     X (P, 256: the block's first layer, e.g. from regnet_interp_affine_f32) -> 256 -> 256 = F (P, 256), the point
 *     feature ScoreNetwork returns -> 512 -> 256 -> 256 -> 128 -> conv_score + bn_score + sigmoid = score (P).
 * Every layer is a bias-free 1x1 convolution + eval BatchNorm (folded to scale/shift) + ReLU.  A wave keeps the
 * activations of 16 points in registers from X to the score (csrc/rowchain.hip); only X is read and only F / score
 * are written -- the layer-wise path moves 15x the bytes.  Same values as six regnet_mlp_layer_f32 calls +
 * regnet_score_head_f32 up to fp32 summation order.
 * `stream_w`: regnet_fp_head_chain_stream_floats() floats = 60 stages of 32 KiB in consumption order -- per layer
 * pair (A: 256 -> M, B: M -> N) and per 128-channel group g of M: four A-stages (rows 128 g + 32 u .. + 32 of W_A, all
 * 256 columns) then N/64 B-stages (rows 64 v .. + 64 of W_B, columns 128 g .. + 128); inside a stage the 16-byte
 * chunk c of row r is stored at chunk position c ^ (r & 15) (low 4 bits) of its row (bank-conflict-free fragment
 * reads).  `affine`: per layer [scale(N) | shift(N)], layers in order: 3328 floats.  Rows and buffers 16-byte aligned.
 * `ticket`: one int32 in device memory, ZERO when the kernel starts (the caller clears it on the same stream): the
 * work queue through which the resident workgroups draw their 128-row blocks.                                          */
int64_t regnet_fp_head_chain_stream_floats(void);
/* A launch hands out the 128-row blocks [block_first, block_first + block_count) of the regnet_fp_head_chain_blocks(P) the
 * rows make (block_count < 0: all of them).  Two launches over disjoint ranges, each with its OWN zeroed ticket word, may
 * run on different streams: the caller can put the last, partial round of blocks (P = 204 800 rows: 1600 blocks = 6 x 256 +
 * 64) beside whatever its main stream runs next instead of leaving 3/4 of the chip idle for a whole pass.            */
int64_t regnet_fp_head_chain_blocks(int64_t P);
/* regnet_fp_head_chain_interp_f32: the same chain with the block's FIRST layer in its prologue -- 3-NN interpolation of
 * the pre-multiplied sparse rows + the narrow skip input + folded BN + ReLU, i.e. regnet_interp_affine_f32's arithmetic
 * (pn2_utils/modules.py:104-131, :500-509) -- so that the (P x 256) first-layer activation is never written:
 * Ys (B, Ns, 256) with element strides ys_sb / ys_sn; idx / dist2 (B*Nd, 3); dense_small (B, Cd <= 4, Nd) with element
 * strides db / dn / dc or NULL; tables = Wd4 (256 x 4) | scale1 (256) | shift1 (256); everything else as above.        */
int regnet_fp_head_chain_interp_f32(const float* Ys, int64_t ys_sb, int64_t ys_sn, const int64_t* idx,
                                    const float* dist2, float eps, const float* dense_small, int64_t db, int64_t dn,
                                    int64_t dc, int64_t Cd_small, const float* tables, int64_t B, int64_t Nd,
                                    const float* stream_w, int64_t n_stages, const float* affine, int64_t affine_floats,
                                    const float* wscore, float score_bias, float score_bn_scale, float score_bn_shift,
                                    float* F, int64_t ldf, float* score, int32_t* ticket, int64_t block_first,
                                    int64_t block_count, void* stream);
int regnet_fp_head_chain_f32(const float* X, int64_t ldx, const float* stream_w, int64_t n_stages,
                             const float* affine, int64_t affine_floats, const float* wscore, float score_bias,
                             float score_bn_scale, float score_bn_shift, float* F, int64_t ldf, float* score,
                             int64_t P, int32_t* ticket, int64_t block_first, int64_t block_count, void* stream);

/* ---- training: the 1x1 convolutions of the shared-MLP blocks on the matrix cores (csrc/tgemm.hip) ------------------
 * nn.Conv1d / nn.Conv2d(kernel_size = 1, bias = False) of pn2_utils/nn/modules/conv.py:20-36, :60-76 under autograd
 * (train.py:376-384), in the tensors' own channel-first layout: X (B, Ci, L), W (Co, Ci), Y (B, Co, L), all contiguous.
 *   fwd    Y[b]  = W . X[b]            dgrad  dX[b] = W^T . dY[b]            wgrad  dW = sum_b dY[b] . X[b]^T
 * regnet_conv1x1_train_supported(Co, Ci, L) != 0: this build runs the shape (channel counts multiples of 16 and >= 32,
 * L a multiple of 4 and >= 64; wgrad additionally L % 16 == 0); otherwise the entry points return REGNET_ERR_SHAPE and
 * the caller keeps its library GEMM.  The weight gradient is split over the point axis into
 * regnet_conv1x1_wgrad_slices() slices per scene whose partial sums (workspace of
 * regnet_conv1x1_wgrad_workspace_bytes() bytes, 16-byte aligned; may be NULL when that is 0) are added in slice
 * order by a second kernel: deterministic.  Same values as the library path up to fp32 summation order.          */
int regnet_conv1x1_train_supported(int64_t Co, int64_t Ci, int64_t L);
int regnet_conv1x1_fwd_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                           void* stream);
int regnet_conv1x1_dgrad_f32(const float* W, const float* dY, float* dX, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                             void* stream);
/* ..._stream: the same two contractions by the persistent kernel (two workgroups per CU draw output tiles from `ticket`,
 * an int32 zeroed by the caller, and keep one LDS ring running across tiles); bit-identical results.                  */
int regnet_conv1x1_fwd_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                  int32_t* ticket, void* stream);
int regnet_conv1x1_dgrad_stream_f32(const float* W, const float* dY, float* dX, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                    int32_t* ticket, void* stream);
/* regnet_conv1x1_fwd_smallci_f32: the same forward for 1 <= Ci <= 8 input channels (the level-1 block's first layer on its
 * grouped rows, 6 -> 128): a store stream -- a thread holds four points of every input channel and writes a float4 per output
 * channel.  L % 4 == 0, X and Y 16-byte aligned, any Co.                                                                */
int regnet_conv1x1_fwd_smallci_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                   void* stream);
/* ..._stats: the same, also leaving sums (2 Co doubles, as regnet_conv1x1_fwd_stats_stream_f32 below) for the BatchNorm that follows.
 * Y = W X is linear in <= 8 inputs: the sums follow from the first and second moments of X (Ci + Ci (Ci + 1) / 2 numbers, fp64
 * above 256 points), Y is not read.  workspace: regnet_conv1x1_smallci_stats_workspace_bytes(B, Ci, L) bytes, 8-byte aligned. */
int64_t regnet_conv1x1_smallci_stats_workspace_bytes(int64_t B, int64_t Ci, int64_t L);
int regnet_conv1x1_fwd_smallci_stats_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                         void* workspace, void* sums, void* stream);
/* ... and their weight gradient: dW = sum over part's first axis, part (regnet_conv1x1_wgrad_smallci_partials(B, Co, L), Co, Ci)
 * written by the call (one partial matrix per scene and slice of the point axis; the caller's sum fixes the order).     */
int64_t regnet_conv1x1_wgrad_smallci_partials(int64_t B, int64_t Co, int64_t L);
int regnet_conv1x1_wgrad_smallci_f32(const float* dY, const float* X, float* part, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                     void* stream);
/* regnet_conv1x1_smallco_f32: the mirror case, 1 <= Co <= 4 output channels (the segmentation head's score convolution, 128 -> 1
 * with bias): dir 0: out (B, Co, L) = W (Co, Ci) . in (B, Ci, L) + bias (Co, may be NULL); dir 1: out (B, Ci, L) = W^T . in
 * (B, Co, L), the input gradient (bias ignored).  L % 4 == 0, in / out 16-byte aligned.  The weight gradient is
 * regnet_conv1x1_wgrad_smallci_f32 with the operands' roles swapped.                                                   */
int regnet_conv1x1_smallco_f32(int dir, const float* W, const float* bias, const float* in, float* out, int64_t B, int64_t Co,
                               int64_t Ci, int64_t L, void* stream);
/* regnet_conv1x1_stream_reserve_slots: the persistent kernels above hold every CU's register file for a whole launch (two
 * workgroups per CU); `slots` of those 2 x CUs workgroup slots stay empty from now on (at most half of them; negative: query
 * only), so that small kernels of ANOTHER stream -- the region stage beside the segmentation head's backward -- find CUs to
 * start on instead of advancing one launch per persistent launch.  Process-wide; returns the previous value.            */
int regnet_conv1x1_stream_reserve_slots(int slots);
/* EXPERIMENT, not on any default path (conv1x1_train.SPLIT_PRODUCTS): the same forward / input gradient with fp32-faithful
 * products on the bf16 matrix pipe -- every operand as the exact sum of three bf16 pieces, six of the nine piece products (the
 * dropped ones are <= 2^-24 relative), fp32 accumulation (csrc/tsplit.hip).  transposed == 0: Y (B, Co, L) = W (Co, Ci) .
 * X (B, Ci, L), optionally on [relu](bscale[i] X[., i, .] + bshift[i]) (NULL: none; Ci <= 1024); transposed == 1: dX (B, Ci, L) =
 * W^T . dY (B, Co, L).  Co, Ci multiples of 16.  workspace: regnet_conv1x1_split_workspace_bytes(Co, Ci, transposed) bytes,
 * 16-byte aligned (the weight's three bf16 planes, rebuilt by every call: the weights move every iteration).             */
/* ... and regnet_sa_chain3_f32 (the level-1 set-abstraction block in one kernel) the same way: csrc/sa_split.hip, fused.SPLIT_PRODUCTS,
 * off.  W2 (128 x 128, row stride ldw2) and W3 (C3 x 128, ldw3) are the packed fp32 weights; `planes`
 * (regnet_sa_chain3_split_plane_bytes(C3) bytes, 16-byte aligned) receives their bf16 pieces when build_planes != 0 and is read as it
 * is otherwise (the caller caches it per weight version).  ticket: a zeroed int32 (work-queue head of the persistent workgroups).
 * group == 64, Cf + 3 <= 8, C3 % 64 == 0.                                                                                  */
int64_t regnet_sa_chain3_split_plane_bytes(int64_t C3);
int regnet_sa_chain3_split_f32(const float* feat, int64_t fb, int64_t fn, int64_t fc, int64_t Cf, const float* xyz, int64_t xb,
                               int64_t xc, int64_t xn, const int64_t* nbr, const int64_t* ctr, const int64_t* count,
                               const int64_t* order, int64_t B, int64_t M, int64_t group,
                               const float* W1, const float* scale1, const float* shift1, const float* W2, int64_t ldw2,
                               const float* scale2, const float* shift2, const float* W3, int64_t ldw3, const float* scale3,
                               const float* shift3, int64_t C3, int relu3, void* planes, int build_planes, float* out, int64_t ldo,
                               int32_t* ticket, void* stream);
int regnet_conv1x1_split_supported(int64_t Co, int64_t Ci, int64_t L);
int64_t regnet_conv1x1_split_workspace_bytes(int64_t Co, int64_t Ci, int64_t transposed);
int regnet_conv1x1_split_f32(int transposed, const float* W, const float* in, float* out, int64_t B, int64_t Co, int64_t Ci,
                             int64_t L, const float* bscale, const float* bshift, int brelu, void* workspace, void* stream);
/* ..._bnrelu: the convolution's input is [relu](scale[i] * X[., i, .] + shift[i]) -- a training BatchNorm (+ ReLU) given as
 * its per-channel affine (regnet_bn_train_stats_f32) -- applied to the operand fragments inside the contraction; X itself
 * is the BatchNorm's INPUT.  regnet_conv1x1_bnrelu_supported: train_supported, Ci <= 512 (the affine table lives in LDS),
 * L % 16 == 0.                                                                                                         */
int regnet_conv1x1_bnrelu_supported(int64_t Co, int64_t Ci, int64_t L);
int regnet_conv1x1_fwd_bnrelu_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                         const float* scale, const float* shift, int relu, int32_t* ticket, void* stream);
int regnet_conv1x1_wgrad_bnrelu_f32(const float* dY, const float* X, float* dW, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                    const float* scale, const float* shift, int relu, void* workspace, void* stream);
/* The forward (plain: scale == NULL; or ..._bnrelu) that also leaves the statistics of the BatchNorm FOLLOWING the convolution:
 * sums (2 Co doubles, 8-byte aligned; zeroed and filled by the call) = per output channel the sum and the sum of squares of Y
 * over all B L points, taken from the output tiles while they are in registers -- nn/modules/conv.py:30-36's bn(conv(x)) without
 * a statistics pass over Y.  Continue with regnet_bn_train_stats_from_sums_f32 / regnet_bn_relu_train_fwd_from_sums_f32.
 * regnet_conv1x1_fwd_stats_supported: train_supported, Co <= 256 (the fp64 table lives in LDS beside the operand ring);
 * affine: Ci <= 256, L % 16 == 0.                                                                                       */
int regnet_conv1x1_fwd_stats_supported(int64_t Co, int64_t Ci, int64_t L, int affine);
int regnet_conv1x1_fwd_stats_stream_f32(const float* W, const float* X, float* Y, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                                        const float* scale, const float* shift, int relu, int32_t* ticket, void* sums,
                                        void* stream);
int64_t regnet_conv1x1_wgrad_slices(int64_t B, int64_t Co, int64_t Ci, int64_t L);
int64_t regnet_conv1x1_wgrad_workspace_bytes(int64_t B, int64_t Co, int64_t Ci, int64_t L);
int regnet_conv1x1_wgrad_f32(const float* dY, const float* X, float* dW, int64_t B, int64_t Co, int64_t Ci, int64_t L,
                             void* workspace, void* stream);

/* ---- host-side numpy-compatible random draws of the region stage (no GPU involved) ------------
 * Replaces the per-centre / per-grasp np.random.choice calls of the reference's Python loops
 * (dataset_utils/get_regiondataset.py:331-337, multi_model/gripper_region_network.py:532-544) while
 * consuming numpy's MT19937 stream identically.  mt_key[624] / mt_pos are np.random.get_state()[1:3]
 * (updated in place; the caller hands them back with np.random.set_state).  counts (rows) int32;
 * out (rows,size) int64 positions; valid (rows) u8 or NULL.  mode 0 = radius groups (n>=size: no
 * replacement, 0<n<size: replacement, n==0: -1), mode 1 = gripper crops (n>size / 5<n<=size / else
 * invalid).                                                                                       */
int regnet_np_choice_rows(uint32_t* mt_key, int32_t* mt_pos, const int32_t* counts, int64_t rows,
                          int64_t size, int mode, int64_t* out, uint8_t* valid);

/* ---- the same draws ON THE DEVICE (round 3): numpy's generator state lives in device memory ------
 * d_mt_key[624] (raw MT19937 words) / d_mt_pos[1] are np.random.get_state()[1:3] in device memory, updated in
 * place by kernels on `stream`; d_counts (rows) int32 device; d_out (rows,size) int64 device; d_valid (rows) u8 device
 * or NULL; semantics, row order and stream consumption exactly as regnet_np_choice_rows.  max_count bounds every
 * count (it sizes the workspace: regnet_np_choice_rows_dev_workspace_ints(rows, max_count) int32 words; the last
 * word is a status flag the caller may read back after the stream has drained: non-zero = a count exceeded
 * max_count and the results are invalid).  No host synchronisation.  Replaces the same reference loops
 * (dataset_utils/get_regiondataset.py:331-337, multi_model/gripper_region_network.py:532-544).            */
int64_t regnet_np_choice_rows_dev_workspace_ints(int64_t rows, int64_t max_count);
int regnet_np_choice_rows_dev(uint32_t* d_mt_key, int32_t* d_mt_pos, const int32_t* d_counts, int64_t rows,
                              int64_t size, int64_t max_count, int mode, int64_t* d_out, uint8_t* d_valid,
                              int32_t* d_workspace, void* stream);

/* regnet_np_rand_doubles_dev: np.random.rand(count) from the device-resident generator (two words per double,
 * randomkit's rk_double), d_out (count) float64 device.  Used by the device-side dataset item (scoredataset.py:52-58). */
int regnet_np_rand_doubles_dev(uint32_t* d_mt_key, int32_t* d_mt_pos, int64_t count, double* d_out, void* stream);

/* regnet_dataset_resample_f32: the gather + colour jitter + tanh of ScoreDataset.__getitem__
 * (dataset_utils/scoredataset.py:60-81) for one record on the device: cloud / color (M,3), score / label (M) float32
 * contiguous; pick (N) int64 rows drawn with numpy's stream; rand6 = the six np.random.rand() values of _noise_color
 * (:52-58: table gains, then the object draws r -> gain 1 - r / 5), float64 device.  Outputs pc (N,6) = [xyz | rgb *
 * gain], tanh(score) (N), label (N).  A pick outside [0, M) ORs 1 into *out_of_range (may be NULL).                 */
int regnet_dataset_resample_f32(const float* cloud, const float* color, const float* score, const float* label, int64_t M,
                                const int64_t* pick, int64_t N, const double* rand6, float* pc, float* score_out,
                                float* label_out, int32_t* out_of_range, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REGNET_HIP_H_ */
