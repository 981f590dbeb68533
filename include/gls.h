/*
 * gls.h -- C ABI of the training-step neighbours of the render path ("Gaussian loss & stats"),
 * what train.py does with the rendered image and the screen-space gradients
 * immediately before / after the rasterizer's backward.
 *
 *   gls_l1_ssim_*            utils/loss_utils.py:17-18 (l1_loss) and :36-63 (ssim / _ssim, 11x11 Gaussian
 *                            window sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) as used by
 *                            train.py:131-132 -- both image statistics in ONE pass over the image pair,
 *                            and d(loss)/d(image) in ONE pass (the reference: 5 depthwise convolutions and
 *                            ~15 elementwise launches forward, twice that backward).
 *   gls_l1_*                 l1_loss alone (BASELINE config 3's loss), same two-pass structure.
 *   gls_densification_stats  train.py:197 (max_radii2D update) + scene/gaussian_model.py:517-519
 *                            (add_densification_stats) for update_filter = radii > 0, in place.
 *
 * Conventions as gsr.h / gab.h: DEVICE pointers, fp32, contiguous; 0 / <0 return codes with
 * gls_last_error(); everything is enqueued on `stream`, nothing synchronises.  Reductions are
 * deterministic (fixed-order partials, no float atomics).
 */
#ifndef GLS_H
#define GLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLS_ABI_VERSION 4
#define GLS_OK 0
#define GLS_E_ARG (-1)
#define GLS_E_HIP (-2)
#define GLS_SSIM_WINDOW 11

int gls_abi_version(void);
const char* gls_last_error(void);

/* floats of scratch (`partial`) the two forward entries need for B images of C x H x W */
int64_t gls_partial_floats(int32_t B, int32_t C, int32_t H, int32_t W);

/* img1, img2: (B,C,H,W).  sums: (B,2) <- scale * { sum |img1-img2| , sum ssim_map } per image (scale = 1/(C*H*W)
 * gives the means the reference returns).  maps: 3*B*C*H*W floats kept for the backward
 * (d map/d mu1, d map/d E[x^2], d map/d E[xy]) or NULL for an evaluation-only call. */
int gls_l1_ssim_forward(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, float scale,
                        float* sums, float* maps, float* partial, void* stream);

/* g: (B,2) DEVICE values dL/d(sums) per image; the same host `scale` as the forward.  d_img1: (B,C,H,W), fully written. */
int gls_l1_ssim_backward(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                         const float* maps, const float* g, float scale, float* d_img1, void* stream);

/* The same with the two upstream gradients as separate DEVICE arrays of B floats each, element i at [i * g_stride] (g_stride 0: one value for
 * every image); either may be NULL (= zero): what autograd hands a node whose two outputs are the scalars themselves -- no (B,2) gradient to
 * assemble, no select / index nodes between the loss arithmetic of train.py:132 and this kernel.  (ABI 2) */
int gls_l1_ssim_backward_split(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                               const float* maps, const float* g_l1, const float* g_ssim, int32_t g_stride, float scale,
                               float* d_img1, void* stream);

/* n elements.  sum: 1 float <- scale * sum |a-b|.  partial: gls_partial_floats(1,1,1,1) floats. */
int gls_l1_forward(int64_t n, const float* a, const float* b, float scale, float* sum, float* partial, void* stream);
/* The same, and d_a: n floats <- scale * sign(a-b) = d(sum)/d(a), written by the pass that reads the pair anyway: a backward whose
 * upstream gradient is the constant 1 (utils/loss_utils.py:17-18 followed by loss.backward(), BASELINE config 3) launches nothing. (ABI 2) */
int gls_l1_forward_grad(int64_t n, const float* a, const float* b, float scale, float* sum, float* partial, float* d_a, void* stream);
/* g: 1 DEVICE float dL/d(sum).  d_a: n floats <- g * scale * sign(a-b). */
int gls_l1_backward(int64_t n, const float* a, const float* b, const float* g, float scale, float* d_a, void* stream);

/* For every splat with radii > 0:  max_radii2D = max(max_radii2D, radii);  xyz_gradient_accum += |viewspace_grad.xy|;
 * denom += 1.   viewspace_grad: (P,3) (the .grad of render()'s viewspace_points); the three state vectors: (P,) / (P,1). */
int gls_densification_stats(int32_t P, const int32_t* radii, const float* viewspace_grad, float* max_radii2D,
                            float* xyz_gradient_accum, float* denom, void* stream);

/* GaussianModel.add_densification_stats(viewspace_point_tensor, update_filter) alone (scene/gaussian_model.py:517-519, called at train.py:198):
 * for every splat with update_filter[i] != 0 (P bytes, a torch.bool tensor):  xyz_gradient_accum += |viewspace_grad[i, :2]|;  denom += 1.
 * viewspace_grad: rows of grad_stride floats (3 for render()'s viewspace_points.grad).  One launch instead of the reference's two masked
 * read-modify-write chains (each a nonzero + gather + scatter with a host sync for the index count).  (ABI 3) */
int gls_add_densification_stats(int32_t P, const uint8_t* update_filter, const float* viewspace_grad, int32_t grad_stride,
                                float* xyz_gradient_accum, float* denom, void* stream);

/* Optional per-kernel timing with hipEvents on the launch stream (what bench.py's roofline.all_kernels / roofline.step read for this library's
 * kernels; libgsr has gsr_profile_*).  Off by default.  When on, every launch of this library is bracketed by an event pair;
 * gls_profile_collect() synchronises the pending pairs, ADDS their elapsed times to a table keyed by kernel name and returns the number of
 * table entries; gls_profile_entry(i, ...) reads entry i (name: a static string such as "gls::k_...", total milliseconds, launches; -1 past the
 * end); gls_profile_reset() empties the table.  (ABI 4) */
int gls_profile_enable(int on);
int gls_profile_collect(void);
int gls_profile_entry(int32_t index, const char** name, double* total_ms, int64_t* launches);
int gls_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* GLS_H */
