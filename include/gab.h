/*
 * gab.h -- C ABI of the MI355X-native FLAME-rigged binding path ("Gaussian-avatar binding").
 *
 * The reference runs this half of the per-frame hot path as ~200 small ATen launches per frame
 * (SURVEY.md 2.1).  Each group below replaces one pure-torch stage of the reference with fused HIP
 * kernels, forward and backward:
 *
 *   gab_flame_*        FlameHead.forward        flame_model/flame.py:485-558  (blend_shapes + lbs,
 *                                               flame_model/lbs.py:25-57,101-195,218-304)
 *   gab_face_frames_*  update_mesh_properties   scene/flame_gaussian_model.py:137-154
 *                      (+ compute_face_orientation utils/graphics_utils.py:116-135 and
 *                       roma.rotmat_to_unitquat / quat_xyzw_to_wxyz)
 *   gab_bind_*         get_xyz / get_scaling / get_rotation   scene/gaussian_model.py:113-150
 *                      (+ roma.quat_product)
 *
 * Conventions: every pointer is a DEVICE pointer; fp32; contiguous row-major tensors with the
 * reference's shapes; index tensors may be int32 or int64 (the reference produces both,
 * SURVEY.md H8) -- pass index_is_i64 accordingly.  Returns 0 / <0 like gsr.h; messages through
 * gab_last_error().  All work is enqueued on `stream`; nothing synchronises.
 */
#ifndef GAB_H
#define GAB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAB_ABI_VERSION 5
#define GAB_OK 0
#define GAB_E_ARG (-1)
#define GAB_E_HIP (-2)

#define GAB_NUM_JOINTS 5            /* FLAME: root, neck, jaw, eye_l, eye_r; parents -1,0,1,1,1 */
#define GAB_POSE_FEATURES 36        /* (J-1)*9 */
/* floats of per-frame workspace written by gab_flame_forward and read by gab_flame_backward */
#define GAB_FLAME_WS_FLOATS 512

/* The FLAME buffers FlameHead registers (flame_model/flame.py:98-129, after add_teeth). */
typedef struct GabRig {
    int32_t V;                 /* vertices (5143)                                   */
    int32_t n_shape;           /* leading betas from `shape` (300)                  */
    int32_t n_expr;            /* trailing betas from `expr` (100)                  */
    const float* v_template;   /* (V,3)                                             */
    const float* shapedirs;    /* (V,3,n_shape+n_expr)                              */
    const float* posedirs;     /* (36, 3V)                                          */
    const float* J_regressor;  /* (5,V) dense                                       */
    const float* lbs_weights;  /* (V,5)                                             */
    int32_t parents[GAB_NUM_JOINTS];  /* host values; parents[0] = -1               */
} GabRig;

int gab_abi_version(void);
const char* gab_last_error(void);

/* ---- FLAME forward / backward (batch 1, zero_centered_at_root_node=False, no landmarks) -------- */
int gab_flame_forward(const GabRig* rig, const float* shape, const float* expr, const float* rotation,
                      const float* neck, const float* jaw, const float* eyes /*6*/, const float* translation,
                      const float* static_offset /*(V,3) or NULL*/,
                      float* verts /*(V,3)*/, float* v_shaped /*(V,3)*/, float* ws /*GAB_FLAME_WS_FLOATS*/, void* stream);

/* d_shape / d_static_offset / dL_dv_shaped may be NULL.  scratch: (V,3) floats.  Every non-NULL output
 * is fully written.  `ws` must be the workspace gab_flame_forward wrote for this frame: it carries the backward's
 * accumulators, zeroed by the forward and zeroed again by this call (so the call may be repeated; no memset).
 * zero_*: up to 8 device buffers (HOST arrays of pointers / float counts, as gab_zero_buffers) that the first backward kernel
 * zero-fills before anything else writes -- the full (T,k) gradient tables whose row t the d_* pointers address; 0 to skip. */
int gab_flame_backward(const GabRig* rig, const float* shape, const float* expr, const float* rotation,
                       const float* neck, const float* jaw, const float* eyes, const float* translation,
                       const float* static_offset, const float* v_shaped, float* ws,
                       const float* dL_dverts, const float* dL_dv_shaped,
                       float* d_shape, float* d_expr, float* d_rotation, float* d_neck, float* d_jaw, float* d_eyes,
                       float* d_translation, float* d_static_offset, float* scratch,
                       int32_t zero_count, float* const* zero_buffers_host, const int32_t* zero_sizes_host, void* stream);

/* ---- FLAME with a PREPARED rig: the per-frame part only (what a training step runs) -----------------------------------------
 * gab_flame_prepare folds everything that does not depend on the per-frame parameters -- the shape block of the blend shapes
 * and static_offset (v_static), the joint regression of that (J_static) and M = J_regressor . shapedirs[:, n_shape:], so that
 * joints = J_static + M . expr (flame_model/lbs.py:222-225 restated: J_regressor (v_template + S beta) is affine in beta) --
 * into `prepared` (gab_flame_prepared_floats(rig) floats).  Call it again whenever shape, static_offset or the rig change.
 * With it the forward is ONE launch (blend shapes of the expression block, chain, skinning) and the backward TWO (skinning; chain
 * and shapedirs^T g side by side), against 3 + 3: same outputs, same `ws` contract as gab_flame_forward / gab_flame_backward
 * (the two backward entries are interchangeable on a state either forward wrote).  The prepared backward produces the gradients
 * of the per-frame parameters only: use gab_flame_backward when shape / static_offset / an external dL/dv_shaped take part. */
int64_t gab_flame_prepared_floats(const GabRig* rig);
int gab_flame_prepare(const GabRig* rig, const float* shape, const float* static_offset /*(V,3) or NULL*/, float* prepared, void* stream);
int gab_flame_forward_prepared(const GabRig* rig, const float* prepared, const float* expr, const float* rotation, const float* neck,
                               const float* jaw, const float* eyes /*6*/, const float* translation,
                               float* verts /*(V,3)*/, float* v_shaped /*(V,3)*/, float* ws /*GAB_FLAME_WS_FLOATS*/, void* stream);
/* ---- a SEQUENCE of frames (render mode: render.py:68-76, fps_benchmark_dataset.py:19-32 walk the timesteps of one avatar under no_grad) ----
 * gab_blend_sequence evaluates the expression block of the blend shapes (flame_model/lbs.py:218-239) for ALL T frames at once,
 *     v_shaped_seq[t][e] = prepared[e] + sum_l shapedirs[e][n_shape + l] * expr_table[t][l],
 * as one (T x n_expr) . (n_expr x 3V) fp32 product on the matrix cores (v_mfma_f32_32x32x2_f32) -- the one GEMM-shaped product of the path
 * (SURVEY.md H9): per frame it is a GEMV over the same 6 MB table.  gab_flame_forward_sequence is gab_flame_forward_prepared for a frame
 * whose row of that table is at hand (v_shaped_row = v_shaped_seq + t * 3V): the expression block is neither read nor multiplied again.
 * Forward only (a frame that needs gradients takes gab_flame_forward_prepared); same outputs as the per-frame entry up to the
 * summation order of the product (<= 1e-6 of the vertices' range, tests/test_binding_gpu.py).                                        */
int gab_blend_sequence(const GabRig* rig, const float* prepared, const float* expr_table /*(T, n_expr)*/, int32_t T,
                       float* v_shaped_seq /*(T, 3V)*/, void* stream);
int gab_flame_forward_sequence(const GabRig* rig, const float* prepared, const float* v_shaped_row /*(3V)*/, const float* expr,
                               const float* rotation, const float* neck, const float* jaw, const float* eyes /*6*/, const float* translation,
                               float* verts /*(V,3)*/, float* v_shaped /*(V,3)*/, float* ws /*GAB_FLAME_WS_FLOATS*/, void* stream);
/* zero_*: as gab_flame_backward, at most 7 buffers (d_expr is zero-filled by the first kernel as well unless one of them covers it) */
int gab_flame_backward_prepared(const GabRig* rig, const float* prepared, const float* rotation, const float* neck, const float* jaw,
                                const float* eyes, const float* v_shaped, float* ws, const float* dL_dverts,
                                float* d_expr, float* d_rotation, float* d_neck, float* d_jaw, float* d_eyes, float* d_translation,
                                float* scratch, int32_t zero_count, float* const* zero_buffers_host, const int32_t* zero_sizes_host,
                                void* stream);

/* ---- select_mesh_by_timestep + update_mesh_properties, backward in TWO launches (prepared rig) -----------------------------
 * Replaces gab_face_frames_backward + gab_flame_backward_prepared (three launches; scene/flame_gaussian_model.py:117-154 under
 * autograd): takes the gradients of the four per-face outputs (any of them NULL), optionally an external dL/d(posed vertices) to
 * add, and writes row t of the six per-frame parameter tables.  vf_begin (V+1) / vf_list (3F x 4 int32): the vertex -> corner
 * table of the topology, one row (4 f + c, i0, i1, i2) per corner c of face f = (i0, i1, i2), rows in vertex order -- static,
 * built once by the caller.  `ws`, scratch (3V floats) and zero_* as gab_flame_backward_prepared. */
int gab_mesh_backward_prepared(const GabRig* rig, const float* prepared, const float* rotation, const float* neck, const float* jaw,
                               const float* eyes, const float* v_shaped, float* ws, const float* verts /*(V,3) posed*/,
                               const int32_t* vf_begin, const int32_t* vf_list, const float* d_center, const float* d_orien_mat,
                               const float* d_scaling, const float* d_orien_quat, const float* dL_dverts /*(V,3) or NULL*/,
                               float* d_expr, float* d_rotation, float* d_neck, float* d_jaw, float* d_eyes, float* d_translation,
                               float* scratch, int32_t zero_count, float* const* zero_buffers_host, const int32_t* zero_sizes_host,
                               void* stream);

/* ---- per-face frames ----------------------------------------------------------------------- */
/* d_verts_zeroed: optional (V,3) buffer the forward zero-fills on the side, to be handed to the backward as its
 * pre-zeroed accumulation target (saves the backward's memset, which costs as much as its kernel); NULL to skip. */
int gab_face_frames_forward(int32_t V, int32_t F, const float* verts, const void* faces /*(F,3)*/, int32_t index_is_i64,
                            float* center /*(F,3)*/, float* orien_mat /*(F,3,3)*/, float* scaling /*(F,1)*/,
                            float* orien_quat /*(F,4) WXYZ*/, float* d_verts_zeroed, void* stream);
/* d_verts (V,3) receives the gradient.  d_verts_is_zero != 0: the caller guarantees it is all-zero already (the buffer the
 * forward prepared); otherwise it is zero-filled first. */
int gab_face_frames_backward(int32_t V, int32_t F, const float* verts, const void* faces, int32_t index_is_i64,
                             const float* d_center, const float* d_orien_mat, const float* d_scaling, const float* d_orien_quat,
                             float* d_verts, int32_t d_verts_is_zero, void* stream);

/* ---- per-splat mesh-local -> world ---------------------------------------------------------- */
/* opacity_logit / out_opacity (N floats each, both or neither): get_opacity = sigmoid(_opacity) (scene/gaussian_model.py:158-160)
 * evaluated by the same launch; the backward entries take out_opacity + d_out_opacity and write d_opacity_logit (all three NULL
 * to skip).  Saves the two elementwise launches the activation costs per frame. */
int gab_bind_forward(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation,
                     const void* binding, int32_t index_is_i64, const float* face_center, const float* face_orien_mat,
                     const float* face_scaling, const float* face_orien_quat,
                     float* out_xyz, float* out_scaling, float* out_rotation,
                     const float* opacity_logit, float* out_opacity, void* stream);
/* d_face: 17*F floats, four contiguous blocks  center (F,3) | orien_mat (F,3,3) | scaling (F,1) | orien_quat (F,4)
 * (so each block is directly the gradient tensor of one face attribute), fully written (zero-filled, then accumulated). */
int gab_bind_backward(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation,
                      const void* binding, int32_t index_is_i64, const float* face_center, const float* face_orien_mat,
                      const float* face_scaling, const float* face_orien_quat,
                      const float* d_out_xyz, const float* d_out_scaling, const float* d_out_rotation,
                      float* d_xyz, float* d_log_scaling, float* d_rotation, float* d_face /*17*F, see above*/,
                      const float* out_opacity, const float* d_out_opacity, float* d_opacity_logit, void* stream);

/* Atomic-free, deterministic variant of gab_bind_backward.  `order` (int32 N): splat indices sorted by face;
 * `face_begin` (int32 F+1): CSR offsets into `order`.  Both depend only on `binding` (build them once per
 * densification step).  Every splat must appear exactly once; d_face needs no zero-fill.
 * Optional two-pass form (all three non-NULL, else all NULL): `splat_face` (int32 N) = binding as int32, `slot` (int32 N)
 * = the inverse permutation of `order` (slot[order[j]] == j), `rows` = caller scratch of GAB_BIND_ROW_FLOATS*N floats.
 * Pass 1 visits the splats in splat order (every per-splat array coalesced) and parks each splat's 17 face
 * contributions as one row at its CSR position, pass 2 sums each face's contiguous rows: same sums in the same order,
 * half the time of the one-pass kernel, whose scattered per-splat accesses in face order bound it. */
#define GAB_BIND_ROW_FLOATS 20
int gab_bind_backward_csr(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation,
                          const float* face_orien_mat, const float* face_scaling, const float* face_orien_quat,
                          const float* d_out_xyz, const float* d_out_scaling, const float* d_out_rotation,
                          const int32_t* order, const int32_t* face_begin,
                          float* d_xyz, float* d_log_scaling, float* d_rotation, float* d_face /*17*F, same four blocks*/,
                          const float* out_opacity, const float* d_out_opacity, float* d_opacity_logit,
                          const int32_t* splat_face, const int32_t* slot, float* rows, void* stream);

/* Utility: zero-fills up to 8 device buffers (sizes in floats) with ONE launch.  `buffers_host` / `sizes_host` are
 * HOST arrays of device pointers / element counts.  Used to build the full (T,k) gradient tables of the per-timestep
 * FLAME parameters around the single row gab_flame_backward writes. */
/* Pass 2 of the two-pass form alone: sums every face's contiguous rows (GAB_BIND_ROW_FLOATS floats each, written at the splats' CSR
 * positions) into d_face (17*F).  For callers whose pass 1 runs elsewhere -- the rasterizer's bound entry writes the rows from its own
 * preprocess backward (include/gsr.h: gsr_backward_bound). */
int gab_bind_backward_faces(int32_t F, const int32_t* face_begin, const float* rows, float* d_face /*17*F*/, void* stream);

int gab_zero_buffers(int32_t count, float* const* buffers_host, const int32_t* sizes_host, void* stream);

/* The frame feed of a RECORDED step (one hipGraph per K frames; scene/flame_gaussian_model.py:117-135 addresses row `timestep` of the
 * flame_param tables with a host integer, which a recording would freeze): copies row schedule[*cursor % n_sched] (schedule NULL: row
 * *cursor) modulo T of `packed` (T x width floats: the per-timestep tables side by side) into `row` (width floats, the static one-row
 * tables the recorded kernels read) and advances the DEVICE cursor by one, modulo the schedule's length (T without a schedule): the
 * cursor stays in [0, length) however long the run.  A kernel like any other: capturable, replays walk the schedule on their own. */
int gab_feed_row(const float* packed, int32_t T, int32_t width, const int32_t* schedule, int32_t n_sched, int32_t* cursor, float* row,
                 void* stream);

/* Optional per-kernel timing with hipEvents on the launch stream (what bench.py's roofline.all_kernels / roofline.step read for this library's
 * kernels; libgsr has gsr_profile_*).  Off by default.  When on, every launch of this library is bracketed by an event pair;
 * gab_profile_collect() synchronises the pending pairs, ADDS their elapsed times to a table keyed by kernel name and returns the number of
 * table entries; gab_profile_entry(i, ...) reads entry i (name: a static string such as "gab::k_...", total milliseconds, launches; -1 past the
 * end); gab_profile_reset() empties the table.  (ABI 5) */
int gab_profile_enable(int on);
int gab_profile_collect(void);
int gab_profile_entry(int32_t index, const char** name, double* total_ms, int64_t* launches);
int gab_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* GAB_H */
