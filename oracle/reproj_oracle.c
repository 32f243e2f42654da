/*
 * oracle/reproj_oracle.c -- CPU restatement of MGNet's self-supervised photometric
 * reprojection loss (forward + hand-derived backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mgnet_amd/ may call, link or import this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * or the reported CPU baseline -- never as the product path.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * tests/golden/reproj_*.npz + kats.npz, which were produced by importing the reference's own
 * mgnet.geometry / mgnet.modeling.loss in the build container (tests/golden/make_golden.py).
 *
 * The restatement is deliberately the *multi-pass, materialise-everything* formulation the
 * reference uses (one array per torch op), so that it is an independent check of the fused
 * single-pass HIP kernels and of their closed-form SSIM/bilinear/pose derivatives.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference
 * repo root).  REAL is float (liboracle_f32.so, the parity oracle: same arithmetic type as the
 * reference, which runs this loss in fp32 -- mg_net.py:827) or double (liboracle_f64.so, used
 * only to judge which of two fp32 results is closer to the truth).
 *
 * Layout: NCHW contiguous, like the reference's tensors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifdef ORACLE_F64
#define SUF(n) n##_f64
#else
#define SUF(n) n##_f32
#endif

#ifdef _OPENMP
#include <omp.h>
#define PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define PAR_FOR
#endif

#define NSLOT 4 /* photometric_losses[i] = [warp_prev, unwarp_prev, warp_next, unwarp_next], loss.py:131-144 */

static REAL r_abs(REAL v) { return v < 0 ? -v : v; }
static REAL r_sign(REAL v) { return (REAL)((v > 0) - (v < 0)); } /* torch abs backward: sign(0)=0 */

int SUF(orc_num_threads)(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------
 * pose_utils.py:9-38 euler2mat : R = Rx(x) . Ry(y) . Rz(z)
 * pose_utils.py:41-51 pose_vec2mat + pose.py:40-46 Pose.from_vec : t = vec[:3], angles = vec[3:]
 * out: R[9] row-major, t[3]
 * ------------------------------------------------------------------------------------- */
static void mat3_mul(const REAL* a, const REAL* b, REAL* o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            o[i * 3 + j] = s;
        }
}
static void euler_mats(const REAL* ang, REAL* xm, REAL* ym, REAL* zm) {
    REAL cx = (REAL)cos(ang[0]), sx = (REAL)sin(ang[0]);
    REAL cy = (REAL)cos(ang[1]), sy = (REAL)sin(ang[1]);
    REAL cz = (REAL)cos(ang[2]), sz = (REAL)sin(ang[2]);
    REAL X[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    REAL Y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    REAL Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    memcpy(xm, X, sizeof X);
    memcpy(ym, Y, sizeof Y);
    memcpy(zm, Z, sizeof Z);
}
void SUF(orc_pose_vec2mat)(const REAL* vec6, REAL* R, REAL* t) {
    REAL xm[9], ym[9], zm[9], xy[9];
    euler_mats(vec6 + 3, xm, ym, zm);
    mat3_mul(xm, ym, xy);
    mat3_mul(xy, zm, R);
    t[0] = vec6[0];
    t[1] = vec6[1];
    t[2] = vec6[2];
}
/* backward of the above: dR[9], dt[3] -> dvec[6] */
static void pose_vec2mat_bwd(const REAL* vec6, const REAL* dR, const REAL* dt, REAL* dvec) {
    REAL xm[9], ym[9], zm[9], xy[9], yz[9], tmp[9];
    euler_mats(vec6 + 3, xm, ym, zm);
    mat3_mul(xm, ym, xy);
    mat3_mul(ym, zm, yz);
    REAL cx = xm[4], sx = xm[7], cy = ym[0], sy = ym[2], cz = zm[0], sz = zm[3];
    /* R = X (Y Z): dX = dR (YZ)^T ; dY = X^T dR Z^T ; dZ = (XY)^T dR */
    REAL dX[9], dY[9], dZ[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL s = 0;
            for (int k = 0; k < 3; ++k) s += dR[i * 3 + k] * yz[j * 3 + k];
            dX[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL s = 0;
            for (int k = 0; k < 3; ++k) s += xm[k * 3 + i] * dR[k * 3 + j];
            tmp[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL s = 0;
            for (int k = 0; k < 3; ++k) s += tmp[i * 3 + k] * zm[j * 3 + k];
            dY[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL s = 0;
            for (int k = 0; k < 3; ++k) s += xy[k * 3 + i] * dR[k * 3 + j];
            dZ[i * 3 + j] = s;
        }
    dvec[0] = dt[0];
    dvec[1] = dt[1];
    dvec[2] = dt[2];
    dvec[3] = dX[4] * (-sx) + dX[5] * (-cx) + dX[7] * cx + dX[8] * (-sx);
    dvec[4] = dY[0] * (-sy) + dY[2] * cy + dY[6] * (-cy) + dY[8] * (-sy);
    dvec[5] = dZ[0] * (-sz) + dZ[1] * (-cz) + dZ[3] * cz + dZ[4] * (-sz);
}

/* camera.py:72-81 Camera.Kinv : a CLONE of K with four entries overwritten (K[0,1], K[1,0],
 * K[2,:] are copied verbatim -- this is not a true inverse when K has skew; parity keeps it). */
void SUF(orc_kinv)(const REAL* K9, REAL* Ki9) {
    memcpy(Ki9, K9, 9 * sizeof(REAL));
    REAL fx = K9[0], fy = K9[4], cx = K9[2], cy = K9[5];
    Ki9[0] = (REAL)1.0 / fx;
    Ki9[4] = (REAL)1.0 / fy;
    Ki9[2] = (REAL)-1.0 * cx / fx;
    Ki9[5] = (REAL)-1.0 * cy / fy;
}

/* depth.py:11-15 inv2depth */
void SUF(orc_inv2depth)(const REAL* inv, REAL* depth, long n) {
    for (long k = 0; k < n; ++k) {
        REAL c = inv[k] < (REAL)1e-6 ? (REAL)1e-6 : inv[k];
        depth[k] = (REAL)1.0 / c;
    }
}

/* ---------------------------------------------------------------------------------------
 * camera_utils.py:24-55 view_synthesis for ONE image b:
 *   camera.py:107-141 reconstruct (target cam, Twc = identity)  P = (Kinv.[u,v,1]) * depth
 *   camera.py:143-182 project on ref cam (Tcw = pose)           Xc = K.(R.P + t); Z=clamp(z,1e-5)
 *                                                              Xn = 2(X/Z)/(W-1)-1, Yn likewise
 *   F.grid_sample(bilinear, zeros, align_corners=True)          ix = ((Xn+1)/2)(W-1)
 * ref: [3,H,W]; depth: [H,W]; outputs warped [3,H,W] and (optionally) per-pixel state.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    REAL ix, iy;    /* sampling position in pixel units (after the padding-mode transform) */
    REAL mx, my;    /* d (transformed position) / d (projected position): 1 for padding_mode "zeros" */
    REAL P[3];      /* back-projected point */
    REAL X, Y, z;   /* K.(R P + t) before the divide */
} WarpState;

/* F.grid_sample padding_mode, align_corners=True (ATen GridSampler.h: reflect_coordinates_set_grad over [0, 2 (size-1)], then
 * clip_coordinates_set_grad): 0 = "zeros" (the position is used as it is; corners outside read 0), 1 = "border", 2 = "reflection" */
static int g_pad_mode = 0;
static REAL pad_coord(REAL in, int size, int mode, REAL* mult) {
    *mult = 1;
    if (mode == 0) return in;
    const REAL hi = (REAL)(size - 1);
    if (mode == 2 && size > 1) {
        /* the reference's CPU path (ATen/native/cpu/GridSamplerKernel.cpp, ComputeLocationBase<align_corners=true>::reflect_coordinates,
         * vectorised): |in| folded modulo 2 (size-1) by a truncated division in working precision -- NOT an exact fmod, which matters for
         * the far-out-of-range positions of points behind the camera -- then min(extra, 2 (size-1) - extra); gradient sign from
         * reflect_coordinates_get_grad: -1 iff (one more flip) xor (in < 0) */
        const REAL ts = 2 * hi;
        const int neg = in < 0;
        const REAL a = in < 0 ? -in : in;
        volatile REAL q = a / ts;                 /* (volatile: each operation rounded to REAL, no fused multiply-subtract) */
        const REAL df = (REAL)trunc((double)q);
        volatile REAL prod = df * ts;
        const REAL extra = a - prod;
        const REAL refl = ts - extra;
        const int flip = extra > refl;
        in = flip ? refl : extra;
        *mult = (flip ^ neg) ? (REAL)-1 : (REAL)1;
    }
    if (in <= 0) { *mult = 0; return 0; }
    if (in >= hi) { *mult = 0; return hi; }
    return in;
}

static void view_synthesis_one(const REAL* ref, const REAL* depth, const REAL* K9, const REAL* R, const REAL* t,
                               int H, int W, REAL* warped, WarpState* st) {
    REAL Ki[9];
    SUF(orc_kinv)(K9, Ki);
    PAR_FOR
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            long p = (long)v * W + u;
            REAL g0 = (REAL)u, g1 = (REAL)v, g2 = 1;
            REAL xn0 = Ki[0] * g0 + Ki[1] * g1 + Ki[2] * g2;
            REAL xn1 = Ki[3] * g0 + Ki[4] * g1 + Ki[5] * g2;
            REAL xn2 = Ki[6] * g0 + Ki[7] * g1 + Ki[8] * g2;
            REAL d = depth[p];
            REAL P0 = xn0 * d, P1 = xn1 * d, P2 = xn2 * d; /* Twc = identity: world == camera frame */
            REAL Q0 = R[0] * P0 + R[1] * P1 + R[2] * P2 + t[0];
            REAL Q1 = R[3] * P0 + R[4] * P1 + R[5] * P2 + t[1];
            REAL Q2 = R[6] * P0 + R[7] * P1 + R[8] * P2 + t[2];
            REAL X = K9[0] * Q0 + K9[1] * Q1 + K9[2] * Q2;
            REAL Y = K9[3] * Q0 + K9[4] * Q1 + K9[5] * Q2;
            REAL z = K9[6] * Q0 + K9[7] * Q1 + K9[8] * Q2;
            REAL Z = z < (REAL)1e-5 ? (REAL)1e-5 : z;
            REAL Xn = 2 * (X / Z) / (REAL)(W - 1) - (REAL)1.0;
            REAL Yn = 2 * (Y / Z) / (REAL)(H - 1) - (REAL)1.0;
            REAL ix = ((Xn + 1) / 2) * (REAL)(W - 1);
            REAL iy = ((Yn + 1) / 2) * (REAL)(H - 1);
            REAL mx, my;
            ix = pad_coord(ix, W, g_pad_mode, &mx);
            iy = pad_coord(iy, H, g_pad_mode, &my);
            REAL fx0 = (REAL)floor(ix), fy0 = (REAL)floor(iy);
            REAL tx = ix - fx0, ty = iy - fy0;
            REAL w00 = (1 - tx) * (1 - ty), w10 = tx * (1 - ty), w01 = (1 - tx) * ty, w11 = tx * ty;
            int in_x0 = fx0 >= 0 && fx0 <= (REAL)(W - 1), in_x1 = fx0 + 1 >= 0 && fx0 + 1 <= (REAL)(W - 1);
            int in_y0 = fy0 >= 0 && fy0 <= (REAL)(H - 1), in_y1 = fy0 + 1 >= 0 && fy0 + 1 <= (REAL)(H - 1);
            long x0 = in_x0 ? (long)fx0 : 0, x1 = in_x1 ? (long)fx0 + 1 : 0;
            long y0 = in_y0 ? (long)fy0 : 0, y1 = in_y1 ? (long)fy0 + 1 : 0;
            for (int c = 0; c < 3; ++c) {
                const REAL* rc = ref + (long)c * H * W;
                REAL acc = 0;
                if (in_x0 && in_y0) acc += rc[y0 * W + x0] * w00;
                if (in_x1 && in_y0) acc += rc[y0 * W + x1] * w10;
                if (in_x0 && in_y1) acc += rc[y1 * W + x0] * w01;
                if (in_x1 && in_y1) acc += rc[y1 * W + x1] * w11;
                warped[(long)c * H * W + p] = acc;
            }
            if (st) {
                WarpState* s = st + p;
                s->ix = ix; s->iy = iy; s->mx = mx; s->my = my;
                s->P[0] = P0; s->P[1] = P1; s->P[2] = P2;
                s->X = X; s->Y = Y; s->z = z;
            }
        }
}

/* stage export used by the golden tests: whole batch, pose given as [B,6] vectors */
void SUF(orc_view_synthesis)(const REAL* ref, const REAL* inv_depth, const REAL* K /*B,3,3*/, const REAL* vec /*B,6*/,
                             int B, int H, int W, REAL* warped) {
    long hw = (long)H * W;
    REAL* depth = (REAL*)malloc(hw * sizeof(REAL));
    for (int b = 0; b < B; ++b) {
        REAL R[9], t[3];
        SUF(orc_pose_vec2mat)(vec + b * 6, R, t);
        SUF(orc_inv2depth)(inv_depth + b * hw, depth, hw);
        view_synthesis_one(ref + b * 3 * hw, depth, K + b * 9, R, t, H, W, warped + b * 3 * hw, NULL);
    }
    free(depth);
}

/* ---------------------------------------------------------------------------------------
 * loss.py:200-220 ssim(x, y, 3, c1=1e-4, c2=9e-4) for one [H,W] channel.
 *   reflect-pad 1 (F.pad 'reflect': index -1 -> 1, H -> H-2), 3x3 avg_pool of x,y,x^2,y^2,xy.
 * out = clamp((1-ssim)/2, 0, 1).
 * ------------------------------------------------------------------------------------- */
static inline int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

#define C1 ((REAL)1e-4)
#define C2 ((REAL)9e-4)

static void ssim_channel(const REAL* x, const REAL* y, int H, int W, REAL* out) {
    PAR_FOR
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            REAL sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int dv = -1; dv <= 1; ++dv)
                for (int du = -1; du <= 1; ++du) {
                    long q = (long)reflect(v + dv, H) * W + reflect(u + du, W);
                    REAL a = x[q], b = y[q];
                    sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
                }
            REAL mu_x = sx / 9, mu_y = sy / 9;
            REAL mu_x_mu_y = mu_x * mu_y, mu_x_sq = mu_x * mu_x, mu_y_sq = mu_y * mu_y;
            REAL sigma_x = sxx / 9 - mu_x_sq, sigma_y = syy / 9 - mu_y_sq, sigma_xy = sxy / 9 - mu_x_mu_y;
            REAL s = (2 * mu_x_mu_y + C1) * (2 * sigma_xy + C2) / ((mu_x_sq + mu_y_sq + C1) * (sigma_x + sigma_y + C2));
            REAL val = ((REAL)1.0 - s) / (REAL)2.0;
            out[(long)v * W + u] = val < 0 ? 0 : (val > 1 ? 1 : val);
        }
}
void SUF(orc_ssim)(const REAL* x, const REAL* y, int N /*B*C*/, int H, int W, REAL* out) {
    for (int n = 0; n < N; ++n) ssim_channel(x + (long)n * H * W, y + (long)n * H * W, H, W, out + (long)n * H * W);
}

/* backward of ssim_channel wrt x, following the autograd graph op by op.
 * dout: d(clamped value) [H,W]  ->  dx [H,W] (accumulated, +=) */
static void ssim_channel_bwd(const REAL* x, const REAL* y, const REAL* dout, int H, int W, REAL* dx) {
    int Hp = H + 2, Wp = W + 2;
    /* per output pixel: gradients wrt the three pooled quantities that depend on x */
    REAL* dmu = (REAL*)calloc((size_t)H * W, sizeof(REAL));   /* d / d mu_x            */
    REAL* dxx = (REAL*)calloc((size_t)H * W, sizeof(REAL));   /* d / d avgpool(x^2)    */
    REAL* dxy = (REAL*)calloc((size_t)H * W, sizeof(REAL));   /* d / d avgpool(x*y)    */
    REAL* dpad = (REAL*)calloc((size_t)Hp * Wp, sizeof(REAL)); /* d / d padded x        */
    PAR_FOR
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            REAL sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int dv = -1; dv <= 1; ++dv)
                for (int du = -1; du <= 1; ++du) {
                    long q = (long)reflect(v + dv, H) * W + reflect(u + du, W);
                    REAL a = x[q], b = y[q];
                    sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
                }
            REAL mu_x = sx / 9, mu_y = sy / 9;
            REAL mxy = mu_x * mu_y, mxs = mu_x * mu_x, mys = mu_y * mu_y;
            REAL sig_x = sxx / 9 - mxs, sig_y = syy / 9 - mys, sig_xy = sxy / 9 - mxy;
            REAL N1 = 2 * mxy + C1, N2 = 2 * sig_xy + C2, D1 = mxs + mys + C1, D2 = sig_x + sig_y + C2;
            REAL s = N1 * N2 / (D1 * D2);
            REAL val = ((REAL)1.0 - s) / (REAL)2.0;
            REAL g = dout[(long)v * W + u];
            REAL dval = (val >= 0 && val <= 1) ? g : 0; /* torch.clamp backward passes at the bounds */
            REAL ds = -dval / 2;
            REAL dN1 = ds * N2 / (D1 * D2), dN2 = ds * N1 / (D1 * D2);
            REAL dD1 = -ds * s / D1, dD2 = -ds * s / D2;
            REAL d_mxy = 2 * dN1, d_sigxy = 2 * dN2, d_mxs = dD1, d_sigx = dD2;
            /* sigma_x = E[x^2] - mu_x^2 ; sigma_xy = E[xy] - mu_x mu_y */
            REAL d_Exx = d_sigx;
            d_mxs -= d_sigx;
            REAL d_Exy = d_sigxy;
            d_mxy -= d_sigxy;
            REAL d_mu = 2 * mu_x * d_mxs + mu_y * d_mxy;
            long p = (long)v * W + u;
            dmu[p] = d_mu; dxx[p] = d_Exx; dxy[p] = d_Exy;
        }
    /* avg_pool2d backward onto the padded image, then x^2 / x*y chain rule (serial: scatter) */
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            long p = (long)v * W + u;
            for (int dv = 0; dv < 3; ++dv)
                for (int du = 0; du < 3; ++du) {
                    int rv = v + dv, ru = u + du; /* padded coords */
                    long q = (long)reflect(rv - 1, H) * W + reflect(ru - 1, W);
                    dpad[(long)rv * Wp + ru] += (dmu[p] + 2 * x[q] * dxx[p] + y[q] * dxy[p]) / 9;
                }
        }
    /* F.pad(reflect) backward: fold the halo back */
    for (int rv = 0; rv < Hp; ++rv)
        for (int ru = 0; ru < Wp; ++ru)
            dx[(long)reflect(rv - 1, H) * W + reflect(ru - 1, W)] += dpad[(long)rv * Wp + ru];
    free(dmu); free(dxx); free(dxy); free(dpad);
}

/* loss.py:169-198 calc_photometric_loss for one image: ssim_w*mean_c(SSIM) + (1-ssim_w)*mean_c(L1) -> [H,W] */
static void photometric_one(const REAL* est, const REAL* img, int H, int W, REAL ssim_w, REAL* out, REAL* scratch) {
    long hw = (long)H * W;
    for (long p = 0; p < hw; ++p) out[p] = 0;
    REAL acc_w = ssim_w, l1_w = (REAL)1 - ssim_w;
    for (int c = 0; c < 3; ++c) {
        ssim_channel(est + c * hw, img + c * hw, H, W, scratch);
        for (long p = 0; p < hw; ++p) out[p] += scratch[p];
    }
    for (long p = 0; p < hw; ++p) {
        REAL l1 = 0;
        for (int c = 0; c < 3; ++c) l1 += r_abs(est[c * hw + p] - img[c * hw + p]);
        out[p] = acc_w * (out[p] / 3) + l1_w * (l1 / 3);
    }
}
void SUF(orc_photometric)(const REAL* est, const REAL* img, int B, int H, int W, REAL ssim_w, REAL* out) {
    long hw = (long)H * W;
    REAL* scratch = (REAL*)malloc(hw * sizeof(REAL));
    for (int b = 0; b < B; ++b) photometric_one(est + b * 3 * hw, img + b * 3 * hw, H, W, ssim_w, out + b * hw, scratch);
    free(scratch);
}

/* loss.py:185,196-197 with ssim_loss_weight == 0: the photometric "map" is the 3-channel L1 map itself, and reduce "min" (loss.py:244:
 * cat along the channels, min over ALL of them) takes the per-pixel minimum over the channels as well as over the sources.  out = min_c,
 * chan = the first minimal channel (torch.min(dim) returns the first minimal index; sources are concatenated in list order, so a
 * per-source first-channel minimum followed by the first-source minimum is the same winner) */
static void l1min_one(const REAL* est, const REAL* img, int H, int W, REAL* out, unsigned char* chan) {
    long hw = (long)H * W;
    for (long p = 0; p < hw; ++p) {
        REAL best = r_abs(est[p] - img[p]);
        int w = 0;
        for (int c = 1; c < 3; ++c) {
            REAL v = r_abs(est[c * hw + p] - img[c * hw + p]);
            if (v < best) { best = v; w = c; }
        }
        out[p] = best;
        chan[p] = (unsigned char)w;
    }
}

/* ---------------------------------------------------------------------------------------
 * loss.py:111-154 MultiViewPhotometricLoss.forward  (+ backward)
 *   automask=True, photometric_reduce_op='min', padding_mode='zeros', n scales, 2 context frames.
 * inputs : inv[n] each [B,1,H,W]; img/prev/nxt [B,3,H,W]; mask [B,1,H,W] bytes or NULL (=all ones,
 *          loss.py:236-237 / :275-276); K [B,3,3]; poses [B,2,6]
 * outputs: losses[2] = {photo_w * L_p, smooth_w * L_s}
 *          if want_grad: d_inv[n] and d_poses = gradient of  g_photo*losses[0] + g_smooth*losses[1]
 *          stage (optional, may be NULL): minmap[n][B,H,W]
 * returns 0, or -1 on bad arguments.
 * ------------------------------------------------------------------------------------- */
static int g_automask = 1, g_reduce_mean = 0;   /* loss.py:139-144 automask_loss, :242-246 photometric_reduce_op (set per call, see below) */

/* options of the next orc_reproj_loss call: automask_loss (1 = reference default), photometric_reduce_op (0 = "min", 1 = "mean";
 * loss.py:105-109 asserts that automasking goes with "min"), padding_mode of the warp (0 "zeros", 1 "border", 2 "reflection") */
int SUF(orc_reproj_options)(int automask, int reduce_mean, int pad_mode) {
    if ((automask != 0 && automask != 1) || (reduce_mean != 0 && reduce_mean != 1) || (automask && reduce_mean) || pad_mode < 0 || pad_mode > 2) return -1;
    g_automask = automask;
    g_reduce_mean = reduce_mean;
    g_pad_mode = pad_mode;   /* camera_utils.py:24-55 view_synthesis(padding_mode=...) -> F.grid_sample */
    return 0;
}

int SUF(orc_reproj_loss)(const REAL* const* inv, int n, const REAL* img, const REAL* prev, const REAL* nxt,
                         const uint8_t* mask, const REAL* K, const REAL* poses, int B, int H, int W,
                         REAL ssim_w, REAL photo_w, REAL smooth_w,
                         REAL* losses, REAL* const* minmap_out,
                         int want_grad, REAL g_photo, REAL g_smooth, REAL* const* d_inv, REAL* d_poses) {
    if (n < 1 || n > 8 || B < 1 || H < 2 || W < 2 || !(ssim_w >= 0)) return -1;
    const int automask = g_automask, reduce_mean = g_reduce_mean;
    /* ssim_loss_weight == 0 (loss.py:196-197): 3-channel L1 maps.  "min": per-pixel minimum over channels and sources (l1min_one).
     * "mean": loss[mask] indexes a [B,3,H,W] map with the [B,1,H,W] mask, which torch refuses (IndexError) -- only the mask-less
     * call exists, and there the mean over the three channels is the ordinary formula with weight 0 */
    const int l1min = (ssim_w == 0) && !reduce_mean;
    if (ssim_w == 0 && (reduce_mean ? mask != NULL : mask == NULL)) return -2;   /* (and "min" without a mask: the default mask is built [B,3,H,W], loss.py:236-237, and cannot index the [B,1,H,W] minimum) */
    const long hw = (long)H * W;
    const REAL* ctx[2] = {prev, nxt};

    /* mask counts (loss.py:245: boolean index then mean over ALL selected pixels of the batch) */
    double n_mask = 0, n_mx = 0, n_my = 0;
    for (int b = 0; b < B; ++b)
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                int m = mask ? mask[b * hw + (long)v * W + u] != 0 : 1;
                n_mask += m;
                if (u < W - 1) n_mx += m; /* mask[:, :, :, :-1] loss.py:284 */
                if (v < H - 1) n_my += m; /* mask[:, :, :-1, :] loss.py:285 */
            }

    REAL* depth = (REAL*)malloc(hw * sizeof(REAL));
    REAL* warped = (REAL*)malloc(2 * 3 * hw * sizeof(REAL)); /* [j][3,H,W] for current (b,i) */
    WarpState* st = (WarpState*)malloc(2 * hw * sizeof(WarpState));
    REAL* pm = (REAL*)malloc(NSLOT * hw * sizeof(REAL));     /* the 4 photometric maps */
    REAL* scratch = (REAL*)malloc(hw * sizeof(REAL));
    REAL* dwarp = (REAL*)malloc(3 * hw * sizeof(REAL));
    REAL* dmap = (REAL*)malloc(hw * sizeof(REAL));
    unsigned char* cwin = (unsigned char*)malloc(NSLOT * hw);   /* l1min: winning channel of each of the 4 maps */
    double photo_sum[8] = {0};
    double dR_acc[2][9], dt_acc[2][3];

    if (want_grad) {
        for (int i = 0; i < n; ++i) memset(d_inv[i], 0, (size_t)B * hw * sizeof(REAL));
        memset(d_poses, 0, (size_t)B * 12 * sizeof(REAL));
    }

    for (int b = 0; b < B; ++b) {
        const REAL* imgb = img + b * 3 * hw;
        REAL R[2][9], t[2][3];
        for (int j = 0; j < 2; ++j) {
            SUF(orc_pose_vec2mat)(poses + (b * 2 + j) * 6, R[j], t[j]); /* loss.py:117-119 */
            /* automask term: unwarped loss, computed once and reused for every scale (loss.py:139-144) */
            if (l1min) l1min_one(ctx[j] + b * 3 * hw, imgb, H, W, pm + (2 * j + 1) * hw, cwin + (2 * j + 1) * hw);
            else photometric_one(ctx[j] + b * 3 * hw, imgb, H, W, ssim_w, pm + (2 * j + 1) * hw, scratch);
            memset(dR_acc[j], 0, sizeof dR_acc[j]);
            memset(dt_acc[j], 0, sizeof dt_acc[j]);
        }
        for (int i = 0; i < n; ++i) {
            SUF(orc_inv2depth)(inv[i] + b * hw, depth, hw); /* loss.py:126 */
            for (int j = 0; j < 2; ++j) {
                view_synthesis_one(ctx[j] + b * 3 * hw, depth, K + b * 9, R[j], t[j], H, W, warped + j * 3 * hw, st + j * hw);
                if (l1min) l1min_one(warped + j * 3 * hw, imgb, H, W, pm + (2 * j) * hw, cwin + (2 * j) * hw);
                else photometric_one(warped + j * 3 * hw, imgb, H, W, ssim_w, pm + (2 * j) * hw, scratch);
            }
            /* loss.py:241-246: cat -> min(dim 1) -> [mask] -> mean.  torch.min(dim) returns the FIRST minimal index */
            double s = 0;
            for (long p = 0; p < hw; ++p) {
                int win = 0;
                REAL best = pm[p];
                if (reduce_mean) {   /* loss.py:242-243: mean over the (warped) maps of their masked means = masked mean of their average */
                    best = (pm[p] + pm[2 * hw + p]) / 2;
                    win = -1;
                } else {
                    for (int k = 1; k < NSLOT; ++k) {
                        if (!automask && (k & 1)) continue;   /* without automasking only the warped maps compete (loss.py:133-144) */
                        if (pm[k * hw + p] < best) { best = pm[k * hw + p]; win = k; }
                    }
                }
                if (minmap_out && minmap_out[i]) minmap_out[i][b * hw + p] = best;
                int m = mask ? mask[b * hw + p] != 0 : 1;
                if (m) s += best;
                dmap[p] = (REAL)win; /* reuse as winner store */
            }
            photo_sum[i] += s;

            if (!want_grad) continue;
            /* ---------------- backward of the photometric part for (b, i) ---------------- */
            for (int j = 0; j < 2; ++j) {
                const REAL* wj = warped + j * 3 * hw;
                const REAL gscale = (REAL)(g_photo * photo_w / ((double)n * n_mask));
                memset(dwarp, 0, 3 * hw * sizeof(REAL));
                for (int c = 0; c < 3; ++c) {
                    /* d loss / d ssim-map_c and L1 */
                    for (long p = 0; p < hw; ++p) {
                        int m = mask ? mask[b * hw + p] != 0 : 1;
                        REAL G = (m && (reduce_mean || (int)dmap[p] == 2 * j)) ? (reduce_mean ? gscale / 2 : gscale) : 0;
                        if (l1min) {   /* the whole gradient goes to the winning channel of the winning map */
                            if (cwin[(2 * j) * hw + p] == c) dwarp[c * hw + p] += G * r_sign(wj[c * hw + p] - imgb[c * hw + p]);
                            continue;
                        }
                        scratch[p] = G * ssim_w / 3;
                        dwarp[c * hw + p] += G * ((REAL)1 - ssim_w) / 3 * r_sign(wj[c * hw + p] - imgb[c * hw + p]);
                    }
                    if (!l1min) ssim_channel_bwd(wj + c * hw, imgb + c * hw, scratch, H, W, dwarp + c * hw);
                }
                /* grid_sample backward wrt the grid (zeros padding, align_corners=True), then
                 * camera.py:170-182 project, pose.py:77-82 transform, camera.py:130-133 reconstruct, depth.py:15 */
                const REAL* refj = ctx[j] + b * 3 * hw;
                const REAL* K9 = K + b * 9;
                REAL Ki[9];
                SUF(orc_kinv)(K9, Ki);
                for (int v = 0; v < H; ++v)
                    for (int u = 0; u < W; ++u) {
                        long p = (long)v * W + u;
                        const WarpState* s_ = st + j * hw + p;
                        REAL fx0 = (REAL)floor(s_->ix), fy0 = (REAL)floor(s_->iy);
                        REAL tx = s_->ix - fx0, ty = s_->iy - fy0;
                        int in_x0 = fx0 >= 0 && fx0 <= (REAL)(W - 1), in_x1 = fx0 + 1 >= 0 && fx0 + 1 <= (REAL)(W - 1);
                        int in_y0 = fy0 >= 0 && fy0 <= (REAL)(H - 1), in_y1 = fy0 + 1 >= 0 && fy0 + 1 <= (REAL)(H - 1);
                        long x0 = in_x0 ? (long)fx0 : 0, x1 = in_x1 ? (long)fx0 + 1 : 0;
                        long y0 = in_y0 ? (long)fy0 : 0, y1 = in_y1 ? (long)fy0 + 1 : 0;
                        REAL gix = 0, giy = 0;
                        for (int c = 0; c < 3; ++c) {
                            const REAL* rc = refj + c * hw;
                            REAL v00 = (in_x0 && in_y0) ? rc[y0 * W + x0] : 0, v10 = (in_x1 && in_y0) ? rc[y0 * W + x1] : 0;
                            REAL v01 = (in_x0 && in_y1) ? rc[y1 * W + x0] : 0, v11 = (in_x1 && in_y1) ? rc[y1 * W + x1] : 0;
                            REAL go = dwarp[c * hw + p];
                            gix += go * ((v10 - v00) * (1 - ty) + (v11 - v01) * ty);
                            giy += go * ((v01 - v00) * (1 - tx) + (v11 - v10) * tx);
                        }
                        gix *= s_->mx; giy *= s_->my;   /* padding-mode transform of the sampling position */
                        /* unnormalise (x (W-1)/2) and normalise (x 2/(W-1)) cancel */
                        REAL dXn = gix * ((REAL)(W - 1) / 2), dYn = giy * ((REAL)(H - 1) / 2);
                        REAL da = dXn * 2 / (REAL)(W - 1), db = dYn * 2 / (REAL)(H - 1); /* d/d(X/Z), d/d(Y/Z) */
                        REAL z = s_->z, Z = z < (REAL)1e-5 ? (REAL)1e-5 : z;
                        REAL dX = da / Z, dY = db / Z;
                        REAL dZ = -(da * s_->X + db * s_->Y) / (Z * Z);
                        REAL dz = (z >= (REAL)1e-5) ? dZ : 0; /* clamp(min) backward */
                        REAL dQ0 = K9[0] * dX + K9[3] * dY + K9[6] * dz;
                        REAL dQ1 = K9[1] * dX + K9[4] * dY + K9[7] * dz;
                        REAL dQ2 = K9[2] * dX + K9[5] * dY + K9[8] * dz;
                        const REAL* Rj = R[j];
                        REAL dP0 = Rj[0] * dQ0 + Rj[3] * dQ1 + Rj[6] * dQ2;
                        REAL dP1 = Rj[1] * dQ0 + Rj[4] * dQ1 + Rj[7] * dQ2;
                        REAL dP2 = Rj[2] * dQ0 + Rj[5] * dQ1 + Rj[8] * dQ2;
                        dt_acc[j][0] += dQ0; dt_acc[j][1] += dQ1; dt_acc[j][2] += dQ2;
                        REAL dQ[3] = {dQ0, dQ1, dQ2};
                        for (int a = 0; a < 3; ++a)
                            for (int c2 = 0; c2 < 3; ++c2) dR_acc[j][a * 3 + c2] += (double)dQ[a] * s_->P[c2];
                        REAL g0 = (REAL)u, g1 = (REAL)v;
                        REAL xn0 = Ki[0] * g0 + Ki[1] * g1 + Ki[2], xn1 = Ki[3] * g0 + Ki[4] * g1 + Ki[5], xn2 = Ki[6] * g0 + Ki[7] * g1 + Ki[8];
                        REAL dd = dP0 * xn0 + dP1 * xn1 + dP2 * xn2;
                        REAL iv = inv[i][b * hw + p];
                        REAL d = depth[p];
                        if (iv >= (REAL)1e-6) d_inv[i][b * hw + p] += -dd * d * d;
                    }
            }
        }
        if (want_grad)
            for (int j = 0; j < 2; ++j) {
                REAL dR[9], dt[3], dv[6];
                for (int k = 0; k < 9; ++k) dR[k] = (REAL)dR_acc[j][k];
                for (int k = 0; k < 3; ++k) dt[k] = (REAL)dt_acc[j][k];
                pose_vec2mat_bwd(poses + (b * 2 + j) * 6, dR, dt, dv);
                for (int k = 0; k < 6; ++k) d_poses[(b * 2 + j) * 6 + k] = dv[k];
            }
    }

    double Lp = 0;
    for (int i = 0; i < n; ++i) Lp += photo_sum[i] / n_mask;
    Lp /= n;

    /* ---- loss.py:257-294 calc_smoothness_loss + depth.py:18-51 calc_smoothness ---- */
    double Ls = 0;
    for (int i = 0; i < n; ++i) {
        double sum_x = 0, sum_y = 0;
        for (int b = 0; b < B; ++b) {
            const REAL* iv = inv[i] + b * hw;
            const REAL* im = img + b * 3 * hw;
            double msum = 0;
            for (long p = 0; p < hw; ++p) msum += iv[p];
            REAL mean = (REAL)(msum / (double)hw);
            REAL mc = mean < (REAL)1e-6 ? (REAL)1e-6 : mean; /* depth.py:48-50 */
            for (long p = 0; p < hw; ++p) scratch[p] = iv[p] / mc;
            REAL* ddn = dmap; /* d / d normalised inverse depth */
            if (want_grad) memset(ddn, 0, hw * sizeof(REAL));
            const REAL cx_ = (REAL)(g_smooth * smooth_w / ((double)n * n_mx * (double)(1 << i)));
            const REAL cy_ = (REAL)(g_smooth * smooth_w / ((double)n * n_my * (double)(1 << i)));
            double bx = 0, by = 0;
            for (int v = 0; v < H; ++v)
                for (int u = 0; u < W; ++u) {
                    long p = (long)v * W + u;
                    int m = mask ? mask[b * hw + p] != 0 : 1;
                    if (u < W - 1) {
                        REAL gx = scratch[p] - scratch[p + 1]; /* image.py:42-54 gradient_x */
                        REAL ig = 0;
                        for (int c = 0; c < 3; ++c) ig += r_abs(im[c * hw + p] - im[c * hw + p + 1]);
                        REAL wx = (REAL)exp(-(ig / 3));
                        REAL sxv = gx * wx;
                        if (m) {
                            bx += r_abs(sxv);
                            if (want_grad) { REAL g = cx_ * r_sign(sxv) * wx; ddn[p] += g; ddn[p + 1] -= g; }
                        }
                    }
                    if (v < H - 1) {
                        REAL gy = scratch[p] - scratch[p + W];
                        REAL ig = 0;
                        for (int c = 0; c < 3; ++c) ig += r_abs(im[c * hw + p] - im[c * hw + p + W]);
                        REAL wy = (REAL)exp(-(ig / 3));
                        REAL syv = gy * wy;
                        if (m) {
                            by += r_abs(syv);
                            if (want_grad) { REAL g = cy_ * r_sign(syv) * wy; ddn[p] += g; ddn[p + W] -= g; }
                        }
                    }
                }
            sum_x += bx; sum_y += by;
            if (want_grad) {
                double dmc = 0;
                for (long p = 0; p < hw; ++p) dmc -= (double)ddn[p] * iv[p] / ((double)mc * mc);
                REAL dmean = (mean >= (REAL)1e-6) ? (REAL)(dmc / (double)hw) : 0;
                for (long p = 0; p < hw; ++p) d_inv[i][b * hw + p] += ddn[p] / mc + dmean;
            }
        }
        Ls += (sum_x / n_mx + sum_y / n_my) / (double)(1 << i);
    }
    Ls /= n;

    losses[0] = (REAL)(Lp * photo_w);
    losses[1] = (REAL)(Ls * smooth_w);
    free(depth); free(warped); free(st); free(pm); free(scratch); free(dwarp); free(dmap); free(cwin);
    return 0;
}

/* depth.py:18-31 calc_smoothness stage export: smoothness_x [B,1,H,W-1], smoothness_y [B,1,H-1,W] */
void SUF(orc_calc_smoothness)(const REAL* inv, const REAL* img, int B, int H, int W, REAL* sx, REAL* sy) {
    long hw = (long)H * W;
    for (int b = 0; b < B; ++b) {
        const REAL* iv = inv + b * hw;
        const REAL* im = img + b * 3 * hw;
        double msum = 0;
        for (long p = 0; p < hw; ++p) msum += iv[p];
        REAL mean = (REAL)(msum / (double)hw);
        REAL mc = mean < (REAL)1e-6 ? (REAL)1e-6 : mean;
        for (int v = 0; v < H; ++v)
            for (int u = 0; u < W; ++u) {
                long p = (long)v * W + u;
                if (u < W - 1) {
                    REAL ig = 0;
                    for (int c = 0; c < 3; ++c) ig += r_abs(im[c * hw + p] - im[c * hw + p + 1]);
                    sx[(long)b * H * (W - 1) + (long)v * (W - 1) + u] = (iv[p] / mc - iv[p + 1] / mc) * (REAL)exp(-(ig / 3));
                }
                if (v < H - 1) {
                    REAL ig = 0;
                    for (int c = 0; c < 3; ++c) ig += r_abs(im[c * hw + p] - im[c * hw + p + W]);
                    sy[(long)b * (H - 1) * W + p] = (iv[p] / mc - iv[p + W] / mc) * (REAL)exp(-(ig / 3));
                }
            }
    }
}
