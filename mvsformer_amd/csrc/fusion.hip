// Depth-map geometric consistency filtering — SURVEY.md §8(f2), the step right after the path: reference
// misc/fusion.py:79-122 (get_reproj -> project_img, vis_filter, ave_fusion) as used by test.py:404-438 (filter_depth).
//
// The reference materializes, per source view, a 3-channel "src pixel -> (x_ref, y_ref, depth_ref)" image, then warps it
// into the reference view with grid_sample, through ~20 [n,h,w,4,1] temporaries and 6 batched 4x4 inverses.  Here one
// lane owns one reference pixel and walks the source views: reference pixel -> source image (with the reference depth),
// the 4 bilinear taps there are back-projected on the fly with THEIR source depths into the reference camera, blended,
// compared against the pixel itself; masks, the averaged depth and the fused 3-D point fall out of the same pass.
// Traffic: (V+1) depth maps read (taps hit L1/L2), outputs written once.  Camera algebra is hoisted into a prep kernel
// (fp64 inverses, one thread per view).
//
// Conventions kept from the reference: pixel centres at +0.5 (get_pixel_grids), homogeneous divides by (w + 1e-9),
// warp coordinates normalized as x/width*2-1, clamped to [-1.1, 1.1], sampled with align_corners=True, zeros padding.
#include "common.h"

namespace {

struct ViewXf {            // per (n, v): 84 floats
    float r2s[16];         // E_src * inv(E_ref)
    float s2r[16];         // E_ref * inv(E_src)
    float Kr[9], Kri[9], Ks[9], Ksi[9];
    float Eri[16];         // inv(E_ref), for the fused world point
};

__device__ void inv4d(const double* A, double* inv) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = A[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); p = r; }
        if (p != c) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double piv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= piv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
}

__device__ void inv3d(const float* K, float* out) {     // K is [4,4] row-major, upper-left 3x3 used
    double m[16] = {0}, mi[16];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i * 4 + j] = K[i * 4 + j];
    m[15] = 1.0;
    inv4d(m, mi);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[i * 3 + j] = (float)mi[i * 4 + j];
}

__global__ void geo_prep_kernel(const float* __restrict__ ref_cam /*[n,2,4,4]*/, const float* __restrict__ src_cam /*[n,v,2,4,4]*/,
                                int n, int v, ViewXf* __restrict__ xf) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * v) return;
    const float* rc = ref_cam + (size_t)(idx / v) * 32;
    const float* sc = src_cam + (size_t)idx * 32;
    double Er[16], Es[16], Eri[16], Esi[16];
    for (int i = 0; i < 16; ++i) { Er[i] = rc[i]; Es[i] = sc[i]; }
    inv4d(Er, Eri);
    inv4d(Es, Esi);
    ViewXf& o = xf[idx];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 4; ++k) { a += Es[i * 4 + k] * Eri[k * 4 + j]; b += Er[i * 4 + k] * Esi[k * 4 + j]; }
            o.r2s[i * 4 + j] = (float)a;
            o.s2r[i * 4 + j] = (float)b;
            o.Eri[i * 4 + j] = (float)Eri[i * 4 + j];
        }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { o.Kr[i * 3 + j] = rc[16 + i * 4 + j]; o.Ks[i * 3 + j] = sc[16 + i * 4 + j]; }
    inv3d(rc + 16, o.Kri);
    inv3d(sc + 16, o.Ksi);
}

struct V3 { float x, y, z; };

// 1/x: v_rcp_f32 (1 ulp) + one Newton step, ~0.5 ulp; shared by the three components of each homogeneous divide
// (the IEEE division sequence per component made the kernel VALU-bound: ~45 divides per pixel per view)
__device__ __forceinline__ float recip(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// pixel (px,py) with depth d in camera A (Kinv) -> camera B coordinates (via M = E_B * inv(E_A)); homogeneous
// divides by (w + 1e-9) kept where the reference has them
__device__ __forceinline__ V3 pix_to_cam(const float* Kinv, const float* M, float px, float py, float d) {
    float cx = Kinv[0] * px + Kinv[1] * py + Kinv[2];
    float cy = Kinv[3] * px + Kinv[4] * py + Kinv[5];
    float cz = Kinv[6] * px + Kinv[7] * py + Kinv[8];
    const float sc = recip(cz + 1e-9f) * d;
    cx *= sc; cy *= sc; cz *= sc;
    float X = M[0] * cx + M[1] * cy + M[2] * cz + M[3];
    float Y = M[4] * cx + M[5] * cy + M[6] * cz + M[7];
    float Z = M[8] * cx + M[9] * cy + M[10] * cz + M[11];
    const float Wh = M[12] * cx + M[13] * cy + M[14] * cz + M[15];
    // idx_cam2world divides by (w+1e-9), idx_world2cam again: both are divisions by ~1
    const float w1 = recip(Wh + 1e-9f);
    X *= w1; Y *= w1; Z *= w1;
    return V3{X, Y, Z};
}

__device__ __forceinline__ V3 cam_to_img(const float* K, V3 c) {      // idx_cam2img: K * c, divided by (z + 1e-9)
    const float ix = K[0] * c.x + K[1] * c.y + K[2] * c.z;
    const float iy = K[3] * c.x + K[4] * c.y + K[5] * c.z;
    const float iz = K[6] * c.x + K[7] * c.y + K[8] * c.z;
    const float den = iz + 1e-9f, zz = recip(den);
    // quotient with one residual correction: correctly rounded in all but rare cases.  A prob-filtered (depth 0) source
    // pixel lands at |x| ~ 1e4 px where a 1-ulp quotient error is already 1e-3 px in the blended coordinate.
    float qx = ix * zz, qy = iy * zz;
    qx = fmaf(fmaf(-den, qx, ix), zz, qx);
    qy = fmaf(fmaf(-den, qy, iy), zz, qy);
    return V3{qx, qy, iz * zz};
}

__global__ __launch_bounds__(256) void geo_filter_kernel(const float* __restrict__ ref_depth, const float* __restrict__ src_depth,
                                                         const float* __restrict__ ref_cam, const ViewXf* __restrict__ xf, int V, int H,
                                                         int W, float dist_thresh, float depth_thresh, float vthresh,
                                                         float* __restrict__ reproj /*[n,v,3,h,w]*/, float* __restrict__ in_range_out,
                                                         float* __restrict__ masks_out /*[n,v,h,w]*/, uint8_t* __restrict__ mask_out,
                                                         float* __restrict__ ave_out, float* __restrict__ points_out /*[n,3,h,w]*/) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, n = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float dref = ref_depth[(size_t)n * HW + pix];
    float msum = 0.0f, dsum = 0.0f;
    for (int v = 0; v < V; ++v) {
        const ViewXf& t = xf[n * V + v];
        const float* sd = src_depth + (size_t)(n * V + v) * HW;
        // project_img(dst = reference): reference pixel -> source image coordinates
        const V3 cs = pix_to_cam(t.Kri, t.r2s, px, py, dref);
        const V3 q = cam_to_img(t.Ks, cs);
        float wx = q.x / (float)W * 2.0f - 1.0f, wy = q.y / (float)H * 2.0f - 1.0f;
        wx = fminf(fmaxf(wx, -1.1f), 1.1f);              // clamp(-1.1, 1.1); NaN propagates like torch.clamp
        wy = fminf(fmaxf(wy, -1.1f), 1.1f);
        const float inr = (-1.0f <= wx && wx <= 1.0f && -1.0f <= wy && wy <= 1.0f) ? 1.0f : 0.0f;
        // grid_sample(align_corners=True, zeros) of the source-grid image srcs2ref_xyd, evaluated tap by tap
        const float ix = ((wx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((wy + 1.0f) / 2.0f) * (float)(H - 1);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx1 = ix - x0f, fx0 = (x0f + 1.0f) - ix, fy1 = iy - y0f, fy0 = (y0f + 1.0f) - iy;   // ATen's tap weights
        float rx = 0.0f, ry = 0.0f, rd = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xs = x0f + (float)(k & 1), ys = y0f + (float)(k >> 1);
            const float wgt = ((k & 1) ? fx1 : fx0) * ((k >> 1) ? fy1 : fy0);
            if (xs >= 0.0f && xs <= (float)(W - 1) && ys >= 0.0f && ys <= (float)(H - 1)) {
                const float ds = sd[(size_t)ys * W + (size_t)xs];
                const V3 cr = pix_to_cam(t.Ksi, t.s2r, xs + 0.5f, ys + 0.5f, ds);
                const V3 im = cam_to_img(t.Kr, cr);
                rx = fmaf(im.x, wgt, rx);
                ry = fmaf(im.y, wgt, ry);
                rd = fmaf(cr.z, wgt, rd);
            }
        }
        const size_t o3 = ((size_t)(n * V + v) * 3) * HW + pix;
        if (reproj) { reproj[o3] = rx; reproj[o3 + HW] = ry; reproj[o3 + 2 * HW] = rd; }
        if (in_range_out) in_range_out[(size_t)(n * V + v) * HW + pix] = inr;
        // vis_filter
        const float ddx = rx - px, ddy = ry - py;
        const float distm = (sqrtf(ddx * ddx + ddy * ddy) < dist_thresh) ? 1.0f : 0.0f;
        const float depm = (fabsf(dref - rd) < fmaxf(dref, rd) * depth_thresh) ? 1.0f : 0.0f;
        const float m = fminf(inr, fminf(distm, depm));
        if (masks_out) masks_out[(size_t)(n * V + v) * HW + pix] = m;
        msum += m;
        dsum += rd * m;
    }
    const float ave = (dsum + dref) / (msum + 1.0f);
    if (mask_out) mask_out[(size_t)n * HW + pix] = (msum >= vthresh - 1.1f) ? 1 : 0;
    if (ave_out) ave_out[(size_t)n * HW + pix] = ave;
    if (points_out) {
        // idx_img2cam(pixel, ave, ref_cam) -> idx_cam2world(ref_cam): world = inv(E_ref) * cam   (test.py:432-434)
        const ViewXf& t = xf[n * V];
        float cx = t.Kri[0] * px + t.Kri[1] * py + t.Kri[2], cy = t.Kri[3] * px + t.Kri[4] * py + t.Kri[5];
        float cz = t.Kri[6] * px + t.Kri[7] * py + t.Kri[8];
        const float sc = recip(cz + 1e-9f) * ave;
        cx *= sc; cy *= sc; cz *= sc;
        const float* M = t.Eri;
        const float wh = recip(M[12] * cx + M[13] * cy + M[14] * cz + M[15] + 1e-9f);
        points_out[((size_t)n * 3 + 0) * HW + pix] = (M[0] * cx + M[1] * cy + M[2] * cz + M[3]) * wh;
        points_out[((size_t)n * 3 + 1) * HW + pix] = (M[4] * cx + M[5] * cy + M[6] * cz + M[7]) * wh;
        points_out[((size_t)n * 3 + 2) * HW + pix] = (M[8] * cx + M[9] * cy + M[10] * cz + M[11]) * wh;
    }
}

// vis_filter (+ ave_fusion) on an already materialized reproj_xyd — the op-level form of fusion.py:101-114.
// masks_in != null: use them (ave_fusion alone); else compute them from in_range and the two thresholds.
__global__ __launch_bounds__(256) void vis_filter_kernel(const float* __restrict__ ref_depth, const float* __restrict__ reproj,
                                                         const float* __restrict__ in_range, const float* __restrict__ masks_in, int V,
                                                         int H, int W, float dist_thresh, float depth_thresh, float vthresh,
                                                         float* __restrict__ masks_out, uint8_t* __restrict__ mask_out,
                                                         float* __restrict__ ave_out) {
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= HW) return;
    const float px = (float)(pix % W) + 0.5f, py = (float)(pix / W) + 0.5f;
    const float dref = ref_depth[(size_t)n * HW + pix];
    float msum = 0.0f, dsum = 0.0f;
    for (int v = 0; v < V; ++v) {
        const size_t o1 = (size_t)(n * V + v) * HW + pix, o3 = (size_t)(n * V + v) * 3 * HW + pix;
        const float rd = reproj[o3 + 2 * HW];
        float m;
        if (masks_in) {
            m = masks_in[o1];
        } else {
            const float ddx = reproj[o3] - px, ddy = reproj[o3 + HW] - py;
            const float distm = (sqrtf(ddx * ddx + ddy * ddy) < dist_thresh) ? 1.0f : 0.0f;
            const float depm = (fabsf(dref - rd) < fmaxf(dref, rd) * depth_thresh) ? 1.0f : 0.0f;
            m = fminf(in_range[o1], fminf(distm, depm));
        }
        if (masks_out) masks_out[o1] = m;
        msum += m;
        dsum += rd * m;
    }
    if (mask_out) mask_out[(size_t)n * HW + pix] = (msum >= vthresh - 1.1f) ? 1 : 0;
    if (ave_out) ave_out[(size_t)n * HW + pix] = (dsum + dref) / (msum + 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------
// Dynamic consistency (fusion.py:116-165 get_reproj_dynamic / vis_filter_dynamic, test.py:475-514): the reference pixel
// is projected into the source view, the SOURCE depth is sampled there (index = pixel coordinate, align_corners=True,
// no clamp), that sample is back-projected into the reference camera.  A view passes at level k in [2, v] when
// dist < k/dist_base and |d_ref - d|/d_ref < k/rel_diff_base; the pixel is kept if for some k at least k views pass.
// Thresholds grow with k, so a view is summarized by the first level it passes (kmin) and the [n,v,v-1,h,w] mask stack
// of the reference is only written on request.
constexpr int kMaxDynViews = 16;

__device__ __forceinline__ int first_level(float rx, float ry, float rd, float px, float py, float dref, int V, float dist_base,
                                           float rel_base) {
    const float ddx = rx - px, ddy = ry - py;
    const float cd = sqrtf(ddx * ddx + ddy * ddy);
    const float dd = fabsf(dref - rd) / dref;
    int kmin = V + 1;
    for (int k = V; k >= 2; --k) {
        const bool ok = (cd < (float)k / dist_base) && (dd < (float)k / rel_base);
        kmin = ok ? k : kmin;
    }
    return kmin;
}

struct DynAcc {
    int cnt[kMaxDynViews + 1];
    float msum, dsum;
    __device__ void init() {
#pragma unroll
        for (int k = 0; k <= kMaxDynViews; ++k) cnt[k] = 0;
        msum = 0.0f;
        dsum = 0.0f;
    }
    __device__ void add(int kmin, float rd, int V) {
#pragma unroll
        for (int k = 2; k <= kMaxDynViews; ++k) cnt[k] += (k >= kmin) ? 1 : 0;
        if (kmin <= V) { msum += 1.0f; dsum += rd; }
    }
    __device__ bool keep(int V) const {
        bool g = false;
#pragma unroll
        for (int k = 2; k <= kMaxDynViews; ++k) g = g || (k <= V && cnt[k] >= k);
        return g;
    }
};

__device__ __forceinline__ void write_levels(uint8_t* masks_out, uint8_t* vis_out, size_t nv, size_t HW, size_t pix, int kmin, int V) {
    if (masks_out)
        for (int k = 2; k <= V; ++k) masks_out[(nv * (V - 1) + (k - 2)) * HW + pix] = (k >= kmin) ? 1 : 0;
    if (vis_out) vis_out[nv * HW + pix] = (kmin <= V) ? 1 : 0;
}

__global__ __launch_bounds__(256) void geo_filter_dynamic_kernel(const float* __restrict__ ref_depth, const float* __restrict__ src_depth,
                                                                 const ViewXf* __restrict__ xf, int V, int H, int W, float dist_base,
                                                                 float rel_base, float* __restrict__ reproj, uint8_t* __restrict__ masks_out,
                                                                 uint8_t* __restrict__ vis_out, uint8_t* __restrict__ geo_out,
                                                                 float* __restrict__ ave_out, float* __restrict__ points_out) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, n = blockIdx.z;
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float dref = ref_depth[(size_t)n * HW + pix];
    const float hx = (float)(W - 1) / 2.0f, hy = (float)(H - 1) / 2.0f;
    DynAcc acc;
    acc.init();
    for (int v = 0; v < V; ++v) {
        const ViewXf& t = xf[n * V + v];
        const float* sd = src_depth + (size_t)(n * V + v) * HW;
        const V3 q = cam_to_img(t.Ks, pix_to_cam(t.Kri, t.r2s, px, py, dref));
        // grid = q/((w-1)/2) - 1, unnormalized by ATen as ((g+1)/2)*(w-1): the sample index is q itself (up to rounding)
        const float ix = ((q.x / hx - 1.0f + 1.0f) / 2.0f) * (float)(W - 1), iy = ((q.y / hy - 1.0f + 1.0f) / 2.0f) * (float)(H - 1);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx1 = ix - x0f, fx0 = (x0f + 1.0f) - ix, fy1 = iy - y0f, fy0 = (y0f + 1.0f) - iy;
        float ds = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xs = x0f + (float)(k & 1), ys = y0f + (float)(k >> 1);
            if (xs >= 0.0f && xs <= (float)(W - 1) && ys >= 0.0f && ys <= (float)(H - 1))
                ds = fmaf(sd[(size_t)ys * W + (size_t)xs], ((k & 1) ? fx1 : fx0) * ((k >> 1) ? fy1 : fy0), ds);
        }
        const V3 cr = pix_to_cam(t.Ksi, t.s2r, q.x, q.y, ds);
        const V3 im = cam_to_img(t.Kr, cr);
        if (reproj) {
            const size_t o3 = ((size_t)(n * V + v) * 3) * HW + pix;
            reproj[o3] = im.x; reproj[o3 + HW] = im.y; reproj[o3 + 2 * HW] = cr.z;
        }
        const int kmin = first_level(im.x, im.y, cr.z, px, py, dref, V, dist_base, rel_base);
        write_levels(masks_out, vis_out, (size_t)(n * V + v), HW, pix, kmin, V);
        acc.add(kmin, cr.z, V);
    }
    const float ave = (acc.dsum + dref) / (acc.msum + 1.0f);
    if (geo_out) geo_out[(size_t)n * HW + pix] = acc.keep(V) ? 1 : 0;
    if (ave_out) ave_out[(size_t)n * HW + pix] = ave;
    if (points_out) {
        const ViewXf& t = xf[n * V];
        float cx = t.Kri[0] * px + t.Kri[1] * py + t.Kri[2], cy = t.Kri[3] * px + t.Kri[4] * py + t.Kri[5];
        float cz = t.Kri[6] * px + t.Kri[7] * py + t.Kri[8];
        const float sc = recip(cz + 1e-9f) * ave;
        cx *= sc; cy *= sc; cz *= sc;
        const float* M = t.Eri;
        const float wh = recip(M[12] * cx + M[13] * cy + M[14] * cz + M[15] + 1e-9f);
        points_out[((size_t)n * 3 + 0) * HW + pix] = (M[0] * cx + M[1] * cy + M[2] * cz + M[3]) * wh;
        points_out[((size_t)n * 3 + 1) * HW + pix] = (M[4] * cx + M[5] * cy + M[6] * cz + M[7]) * wh;
        points_out[((size_t)n * 3 + 2) * HW + pix] = (M[8] * cx + M[9] * cy + M[10] * cz + M[11]) * wh;
    }
}

// op-level vis_filter_dynamic (fusion.py:153-165) on a materialized reproj_xyd, plus the reduction of test.py:503-511
__global__ __launch_bounds__(256) void vis_filter_dynamic_kernel(const float* __restrict__ ref_depth, const float* __restrict__ reproj, int V,
                                                                 int H, int W, float dist_base, float rel_base,
                                                                 uint8_t* __restrict__ masks_out, uint8_t* __restrict__ vis_out,
                                                                 uint8_t* __restrict__ geo_out, float* __restrict__ ave_out) {
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= HW) return;
    const float px = (float)(pix % W) + 0.5f, py = (float)(pix / W) + 0.5f;
    const float dref = ref_depth[(size_t)n * HW + pix];
    DynAcc acc;
    acc.init();
    for (int v = 0; v < V; ++v) {
        const size_t o3 = (size_t)(n * V + v) * 3 * HW + pix;
        const float rd = reproj[o3 + 2 * HW];
        const int kmin = first_level(reproj[o3], reproj[o3 + HW], rd, px, py, dref, V, dist_base, rel_base);
        write_levels(masks_out, vis_out, (size_t)(n * V + v), HW, pix, kmin, V);
        acc.add(kmin, rd, V);
    }
    if (geo_out) geo_out[(size_t)n * HW + pix] = acc.keep(V) ? 1 : 0;
    if (ave_out) ave_out[(size_t)n * HW + pix] = (acc.dsum + dref) / (acc.msum + 1.0f);
}

// prob_filter, fusion.py:69-77: AND over the C confidence channels of (conf[:, i] > thresh[i]); optionally zeroes a
// depth map in place where the test fails (test.py:414-418, `src_depths[:, ids] *= mask`).
__global__ __launch_bounds__(256) void prob_filter_kernel(const float* __restrict__ conf, int C, size_t HW, float t0, float t1, float t2,
                                                          float t3, uint8_t* __restrict__ mask, float* __restrict__ depth) {
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= HW) return;
    const float th[4] = {t0, t1, t2, t3};
    bool keep = true;
    for (int c = 0; c < C; ++c) keep = keep && (conf[((size_t)n * C + c) * HW + pix] > th[c]);
    if (mask) mask[(size_t)n * HW + pix] = keep ? 1 : 0;
    if (depth) depth[(size_t)n * HW + pix] *= keep ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int mvs_vis_filter_fwd(const float* ref_depth, const float* reproj_xyd, const float* in_range, const float* masks_in, int n,
                                  int v, int H, int W, float img_dist_thresh, float depth_thresh, float vthresh, float* masks,
                                  uint8_t* mask, float* ref_depth_ave, mvs_stream_t stream) {
    MVS_REQUIRE(ref_depth && reproj_xyd && (in_range || masks_in), "mvs_vis_filter_fwd: null pointer");
    MVS_REQUIRE(n >= 1 && n <= 65535 && v >= 1 && H >= 1 && W >= 1, "mvs_vis_filter_fwd: bad shape n=%d v=%d H=%d W=%d", n, v, H, W);
    dim3 grid((unsigned)mvs::ceil_div((long long)H * W, 256LL), n);
    hipLaunchKernelGGL(vis_filter_kernel, grid, dim3(256), 0, MVS_STREAM(stream), ref_depth, reproj_xyd, in_range, masks_in, v, H, W,
                       img_dist_thresh, depth_thresh, vthresh, masks, mask, ref_depth_ave);
    return mvs::finish_launch("mvs_vis_filter_fwd");
}

extern "C" int mvs_prob_filter(const float* conf, int n, int C, int64_t HW, const float* thresh_host, uint8_t* mask, float* depth_inplace,
                               mvs_stream_t stream) {
    MVS_REQUIRE(conf && thresh_host && (mask || depth_inplace), "mvs_prob_filter: null pointer");
    MVS_REQUIRE(n >= 1 && n <= 65535 && C >= 1 && C <= 4 && HW >= 1, "mvs_prob_filter: bad shape n=%d C=%d HW=%lld", n, C, (long long)HW);
    float t[4] = {0, 0, 0, 0};
    for (int c = 0; c < C; ++c) t[c] = thresh_host[c];
    dim3 grid((unsigned)mvs::ceil_div((long long)HW, 256LL), n);
    hipLaunchKernelGGL(prob_filter_kernel, grid, dim3(256), 0, MVS_STREAM(stream), conf, C, (size_t)HW, t[0], t[1], t[2], t[3], mask,
                       depth_inplace);
    return mvs::finish_launch("mvs_prob_filter");
}

extern "C" int mvs_geo_filter_dynamic_fwd(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams,
                                          int n, int v, int H, int W, float dist_base, float rel_diff_base, void* workspace,
                                          float* reproj_xyd, uint8_t* masks, uint8_t* vis_mask, uint8_t* geo_mask, float* ref_depth_ave,
                                          float* points, mvs_stream_t stream) {
    MVS_REQUIRE(ref_depth && src_depths && ref_cam && src_cams && workspace, "mvs_geo_filter_dynamic_fwd: null pointer");
    MVS_REQUIRE(n >= 1 && n <= 65535 && v >= 2 && v <= kMaxDynViews && H >= 2 && W >= 2,
                "mvs_geo_filter_dynamic_fwd: bad shape n=%d v=%d (2..%d) H=%d W=%d", n, v, kMaxDynViews, H, W);
    MVS_REQUIRE(dist_base > 0.0f && rel_diff_base > 0.0f, "mvs_geo_filter_dynamic_fwd: bases must be positive");
    hipStream_t s = MVS_STREAM(stream);
    ViewXf* xf = reinterpret_cast<ViewXf*>(workspace);
    hipLaunchKernelGGL(geo_prep_kernel, dim3(mvs::ceil_div(n * v, 64)), dim3(64), 0, s, ref_cam, src_cams, n, v, xf);
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), n), block(64, 4);
    hipLaunchKernelGGL(geo_filter_dynamic_kernel, grid, block, 0, s, ref_depth, src_depths, xf, v, H, W, dist_base, rel_diff_base,
                       reproj_xyd, masks, vis_mask, geo_mask, ref_depth_ave, points);
    return mvs::finish_launch("mvs_geo_filter_dynamic_fwd");
}

extern "C" int mvs_vis_filter_dynamic_fwd(const float* ref_depth, const float* reproj_xyd, int n, int v, int H, int W, float dist_base,
                                          float rel_diff_base, uint8_t* masks, uint8_t* vis_mask, uint8_t* geo_mask, float* ref_depth_ave,
                                          mvs_stream_t stream) {
    MVS_REQUIRE(ref_depth && reproj_xyd, "mvs_vis_filter_dynamic_fwd: null pointer");
    MVS_REQUIRE(n >= 1 && n <= 65535 && v >= 2 && v <= kMaxDynViews && H >= 1 && W >= 1,
                "mvs_vis_filter_dynamic_fwd: bad shape n=%d v=%d (2..%d) H=%d W=%d", n, v, kMaxDynViews, H, W);
    MVS_REQUIRE(dist_base > 0.0f && rel_diff_base > 0.0f, "mvs_vis_filter_dynamic_fwd: bases must be positive");
    dim3 grid((unsigned)mvs::ceil_div((long long)H * W, 256LL), n);
    hipLaunchKernelGGL(vis_filter_dynamic_kernel, grid, dim3(256), 0, MVS_STREAM(stream), ref_depth, reproj_xyd, v, H, W, dist_base,
                       rel_diff_base, masks, vis_mask, geo_mask, ref_depth_ave);
    return mvs::finish_launch("mvs_vis_filter_dynamic_fwd");
}

extern "C" int64_t mvs_geo_filter_workspace_bytes(int n, int v) { return (int64_t)n * v * (int64_t)sizeof(ViewXf); }

extern "C" int mvs_geo_filter_fwd(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams, int n,
                                  int v, int H, int W, float img_dist_thresh, float depth_thresh, float vthresh, void* workspace,
                                  float* reproj_xyd, float* in_range, float* masks, uint8_t* mask, float* ref_depth_ave, float* points,
                                  mvs_stream_t stream) {
    MVS_REQUIRE(ref_depth && src_depths && ref_cam && src_cams && workspace, "mvs_geo_filter_fwd: null pointer");
    MVS_REQUIRE(n >= 1 && n <= 65535 && v >= 1 && H >= 2 && W >= 2, "mvs_geo_filter_fwd: bad shape n=%d v=%d H=%d W=%d", n, v, H, W);
    hipStream_t s = MVS_STREAM(stream);
    ViewXf* xf = reinterpret_cast<ViewXf*>(workspace);
    hipLaunchKernelGGL(geo_prep_kernel, dim3(mvs::ceil_div(n * v, 64)), dim3(64), 0, s, ref_cam, src_cams, n, v, xf);
    dim3 grid(mvs::ceil_div(W, 64), mvs::ceil_div(H, 4), n), block(64, 4);
    hipLaunchKernelGGL(geo_filter_kernel, grid, block, 0, s, ref_depth, src_depths, ref_cam, xf, v, H, W, img_dist_thresh, depth_thresh,
                       vthresh, reproj_xyd, in_range, masks, mask, ref_depth_ave, points);
    return mvs::finish_launch("mvs_geo_filter_fwd");
}
