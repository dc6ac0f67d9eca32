// MTCNN face detector kernels: the landmark detector in front of the align-and-crop step,
// /root/reference/utils/detect_align_crop_data.py:18-20 (`MTCNN(select_largest=True, post_process=False)`) and :99
// (`mtcnn.detect(img_batch, landmarks=True)`) — facenet_pytorch's P-Net / R-Net / O-Net cascade.
//
// The three networks are tiny (3..128 channels, 12x12 .. 48x48 crops, one image pyramid): a few hundred MFLOP per image, no
// channel count a multiple of 64 — not MFMA work.  They run in fp32 on the VALU with coalesced NCHW accesses so that the
// thresholded decisions of the cascade (0.6 / 0.7 / 0.7) agree with the fp32 oracle; everything data-dependent and small
// (box generation, NMS, regression, cropping windows) is host logic in faceposegenerator_amd/mtcnn.py, as upstream does it.
#include "idb_common.h"

namespace {

// crop + F.interpolate(mode="area") (= adaptive average pooling) + (x - sub) * mul, uint8 NHWC in -> fp32 NCHW out.
// boxes: [n][5] = {image, y0, y1, x0, x1} (y1 / x1 exclusive, already clipped to the image)
__global__ __launch_bounds__(256) void crop_resize_area_kernel(const uint8_t* src, int h, int w, int c, const int* boxes, int n, float* out,
                                                               int oh, int ow, float sub, float mul) {
    const long long total = (long long)n * c * oh * ow;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % ow);
    const int oy = (int)((idx / ow) % oh);
    const int ch = (int)((idx / ((long long)ow * oh)) % c);
    const int k = (int)(idx / ((long long)ow * oh * c));
    const int img = boxes[k * 5], y0 = boxes[k * 5 + 1], y1 = boxes[k * 5 + 2], x0 = boxes[k * 5 + 3], x1 = boxes[k * 5 + 4];
    const int ih = y1 - y0, iw = x1 - x0;
    // adaptive pooling window of output (oy, ox): [floor(o * in / out), ceil((o + 1) * in / out))
    const int ys = (int)(((long long)oy * ih) / oh), ye = (int)((((long long)(oy + 1)) * ih + oh - 1) / oh);
    const int xs = (int)(((long long)ox * iw) / ow), xe = (int)((((long long)(ox + 1)) * iw + ow - 1) / ow);
    float s = 0.f;
    for (int y = ys; y < ye; ++y) {
        const uint8_t* row = src + (((long long)img * h + (y0 + y)) * w + x0) * c + ch;
        for (int x = xs; x < xe; ++x) s += (float)row[(long long)x * c];
    }
    out[idx] = (s / (float)((ye - ys) * (xe - xs)) - sub) * mul;
}

// valid convolution, stride 1, fp32 NCHW, + bias, + PReLU (per output channel) when slope != nullptr
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const float* x, const float* wt, const float* bias, const float* slope, float* y,
                                                         int batch, int cin, int h, int w, int cout, int kh, int kw) {
    const int oh = h - kh + 1, ow = w - kw + 1;
    const long long total = (long long)batch * cout * oh * ow;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % ow);
    const int oy = (int)((idx / ow) % oh);
    const int co = (int)((idx / ((long long)ow * oh)) % cout);
    const int b = (int)(idx / ((long long)ow * oh * cout));
    float acc = bias ? bias[co] : 0.f;
    const float* wp = wt + (long long)co * cin * kh * kw;
    for (int ci = 0; ci < cin; ++ci) {
        const float* xp = x + (((long long)b * cin + ci) * h + oy) * w + ox;
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx) acc = fmaf(xp[ky * w + kx], wp[(ci * kh + ky) * kw + kx], acc);
    }
    if (slope) acc = acc >= 0.f ? acc : acc * slope[co];
    y[idx] = acc;
}

// nn.MaxPool2d(k, stride, ceil_mode=True), fp32, planes = batch * channels
__global__ __launch_bounds__(256) void maxpool2d_f32_kernel(const float* x, float* y, int planes, int h, int w, int k, int stride, int oh, int ow) {
    const long long total = (long long)planes * oh * ow;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % ow);
    const int oy = (int)((idx / ow) % oh);
    const long long pl = idx / ((long long)ow * oh);
    const float* xp = x + pl * h * w;
    float m = -INFINITY;
    for (int ky = 0; ky < k; ++ky) {
        const int yy = oy * stride + ky;
        if (yy >= h) break;
        for (int kx = 0; kx < k; ++kx) {
            const int xx = ox * stride + kx;
            if (xx >= w) break;
            m = fmaxf(m, xp[yy * w + xx]);
        }
    }
    y[idx] = m;
}

// softmax over a channel pair: p1 = softmax([a0, a1])[1] for x [batch][2][hw]
__global__ __launch_bounds__(256) void softmax_pairs_kernel(const float* x, float* p1, int batch, int hw) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)batch * hw) return;
    const int b = (int)(idx / hw), i = (int)(idx - (long long)b * hw);
    const float a0 = x[((long long)b * 2) * hw + i], a1 = x[((long long)b * 2 + 1) * hw + i];
    const float mx = fmaxf(a0, a1);
    const float e0 = expf(a0 - mx), e1 = expf(a1 - mx);
    p1[idx] = e1 / (e0 + e1);
}

inline int pool_out(int n, int k, int s) {              // ceil_mode, padding 0 (the last window must start inside the input)
    int o = (n - k + s - 1) / s + 1;
    if ((o - 1) * s >= n) --o;
    return o < 1 ? 1 : o;
}

// Suppression bit matrix of greedy NMS (facenet_pytorch detect_face.py: torchvision batched_nms "Union", nms_numpy "Min"): boxes come
// sorted by descending score; bit j of mask[i][j / 64] says "box i, if kept, removes box j" (j > i, same image, overlap not <= thr:
// a NaN overlap of two degenerate boxes removes under "Min" and keeps under "Union", as mtcnn._nms does on the host).  fp32 arithmetic in exactly the host
// routine's operation order with contraction off (__f*_rn), so the host scan over these words keeps the same boxes as
// mtcnn._nms.  One wave per 64 x 64 block, the column boxes through LDS.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ image, int n, float thr,
                                                      int use_min, float one, unsigned long long* __restrict__ mask, int words) {
    __shared__ float cb[64][5];
    __shared__ int ci[64];
    const int bw = blockIdx.x, bi = blockIdx.y, t = threadIdx.x;
    const int i = bi * 64 + t, j0 = bw * 64;
    if (bw < bi) {                                        // every column index is below every row index
        if (i < n) mask[(long long)i * words + bw] = 0ull;
        return;
    }
    const int jc = min(j0 + t, n - 1);
    const float cx1 = boxes[jc * 4 + 0], cy1 = boxes[jc * 4 + 1], cx2 = boxes[jc * 4 + 2], cy2 = boxes[jc * 4 + 3];
    cb[t][0] = cx1; cb[t][1] = cy1; cb[t][2] = cx2; cb[t][3] = cy2;
    cb[t][4] = __fmul_rn(__fadd_rn(__fsub_rn(cx2, cx1), one), __fadd_rn(__fsub_rn(cy2, cy1), one));
    ci[t] = image ? image[jc] : 0;
    __syncthreads();
    if (i >= n) return;
    const float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
    const float area = __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), one), __fadd_rn(__fsub_rn(y2, y1), one));
    const int im = image ? image[i] : 0;
    unsigned long long bits = 0ull;
    for (int k = 0; k < 64; ++k) {
        const int j = j0 + k;
        if (j <= i || j >= n || ci[k] != im) continue;
        const float w = fmaxf(0.f, __fadd_rn(__fsub_rn(fminf(x2, cb[k][2]), fmaxf(x1, cb[k][0])), one));
        const float h = fmaxf(0.f, __fadd_rn(__fsub_rn(fminf(y2, cb[k][3]), fmaxf(y1, cb[k][1])), one));
        const float inter = __fmul_rn(w, h);
        const float o = use_min ? __fdiv_rn(inter, fminf(area, cb[k][4])) : __fdiv_rn(inter, __fsub_rn(__fadd_rn(area, cb[k][4]), inter));
        if (use_min ? !(o <= thr) : (o > thr)) bits |= 1ull << k;      // NaN (0 / 0 of degenerate boxes): Min drops, Union keeps (mtcnn._nms)
    }
    mask[(long long)i * words + bw] = bits;
}

}  // namespace

extern "C" int idb_crop_resize_area_u8(const uint8_t* src, int32_t batch, int32_t h, int32_t w, int32_t channels, const int32_t* boxes,
                                       int32_t n, float* out, int32_t out_h, int32_t out_w, float sub, float mul, void* stream) {
    IDB_REQUIRE(src && boxes && out && batch > 0 && h > 0 && w > 0 && channels > 0 && n >= 0 && out_h > 0 && out_w > 0,
                "idb_crop_resize_area_u8: bad arguments");
    if (n == 0) return IDB_OK;
    const long long total = (long long)n * channels * out_h * out_w;
    IDB_REQUIRE(total < (1LL << 31) * 256, "idb_crop_resize_area_u8: too many outputs");
    hipLaunchKernelGGL(crop_resize_area_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, h, w, channels,
                       boxes, n, out, out_h, out_w, sub, mul);
    IDB_CHECK_LAUNCH("idb_crop_resize_area_u8");
    return IDB_OK;
}

extern "C" int idb_conv2d_f32(const float* x, const float* w, const float* bias, const float* prelu, float* y, int32_t batch, int32_t cin,
                              int32_t h, int32_t w_, int32_t cout, int32_t kh, int32_t kw, void* stream) {
    IDB_REQUIRE(x && w && y && batch > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && h >= kh && w_ >= kw, "idb_conv2d_f32: bad arguments");
    const long long total = (long long)batch * cout * (h - kh + 1) * (w_ - kw + 1);
    IDB_REQUIRE(total < (1LL << 31) * 256, "idb_conv2d_f32: too many outputs");
    hipLaunchKernelGGL(conv2d_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, prelu, y, batch,
                       cin, h, w_, cout, kh, kw);
    IDB_CHECK_LAUNCH("idb_conv2d_f32");
    return IDB_OK;
}

extern "C" int idb_maxpool2d_f32(const float* x, float* y, int32_t planes, int32_t h, int32_t w, int32_t k, int32_t stride, void* stream) {
    IDB_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && k > 0 && stride > 0, "idb_maxpool2d_f32: bad arguments");
    const int oh = pool_out(h, k, stride), ow = pool_out(w, k, stride);
    const long long total = (long long)planes * oh * ow;
    hipLaunchKernelGGL(maxpool2d_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, planes, h, w, k,
                       stride, oh, ow);
    IDB_CHECK_LAUNCH("idb_maxpool2d_f32");
    return IDB_OK;
}

extern "C" int idb_softmax_pairs_f32(const float* x, float* p1, int32_t batch, int32_t hw, void* stream) {
    IDB_REQUIRE(x && p1 && batch > 0 && hw > 0, "idb_softmax_pairs_f32: bad arguments");
    const long long total = (long long)batch * hw;
    hipLaunchKernelGGL(softmax_pairs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, p1, batch, hw);
    IDB_CHECK_LAUNCH("idb_softmax_pairs_f32");
    return IDB_OK;
}

extern "C" int idb_nms_mask(const float* boxes, const int32_t* image, int32_t n, float thr, int32_t use_min, int32_t plus_one, uint64_t* mask,
                            void* stream) {
    IDB_REQUIRE(boxes && mask && n > 0 && n <= (1 << 20), "idb_nms_mask: bad arguments");
    const int words = (n + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, (hipStream_t)stream, boxes, image, n, thr, use_min, plus_one ? 1.f : 0.f,
                       (unsigned long long*)mask, words);
    IDB_CHECK_LAUNCH("idb_nms_mask");
    return IDB_OK;
}
