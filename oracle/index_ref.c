/* Oracle, plain C: the integer / index arithmetic of the hot path restated without any tensor library, so that the
 * bit-exact requirements (Grid-Pool frame indices, Interp1d knot indices and values) are pinned by code that shares
 * nothing with torch or with the HIP kernels.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench.py's
 * cpu_baseline leg); never linked into the product.  Compile with -ffp-contract=off -fno-fast-math (see
 * __graft_entry__.build_oracle): every operation below must round exactly once, in this order.
 *
 * Follows (file:line under /root/reference, and ATen semantics of the torch calls made there):
 *   grid_time_index : x3d_coarse.py:394-403 (cdf -> grid = (cdf-0.5)*2, F.grid_sample(align_corners=True)):
 *                     ATen grid_sampler_unnormalize: i_t = ((coord+1)/2)*(T-1), i0 = floor(i_t), w1 = i_t - i0
 *   interp1d        : interp1d.py:100-141: ind = searchsorted(x, xnew) (left) - 1 clamped to [0, N-2];
 *                     slope = (y[i+1]-y[i]) / (eps + (x[i+1]-x[i])); ynew = y[i] + slope * (xnew - x[i])
 *   resize_index    : train_fine.py:199 / train_coarse_fineFEAT.py:226 F.interpolate(mode='linear'):
 *                     ATen area_pixel_compute_source_index (align_corners: j*(Tin-1)/(L-1); else the half-pixel form,
 *                     which the CPU build contracts into one FMA), i0 = (int)src, lambda = src - i0
 */
#include <math.h>
#include <stdint.h>

void cfn_ref_grid_time_index(const float* cdf, int n, int T, int32_t* i0, float* w1) {
    for (int k = 0; k < n; ++k) {
        const float coord = (cdf[k] - 0.5f) * 2.0f;
        const float it = ((coord + 1.0f) / 2.0f) * (float)(T - 1);
        const float fl = floorf(it);
        i0[k] = (int32_t)fl;
        w1[k] = it - fl;
    }
}

/* x, y: (B, N) rows (xrow / yrow = 0 broadcasts the single row); xnew: (B, P); outputs (B, P) */
void cfn_ref_interp1d(const float* x, const float* y, const float* xnew, float* ynew, int64_t* ind, int B, int N, int P,
                      int xrow, int yrow, int qrow) {
    const float eps = 1.1920928955078125e-07f;   /* torch.finfo(torch.float32).eps */
    for (int b = 0; b < B; ++b) {
        const float* xb = x + (xrow ? (long)b * N : 0);
        const float* yb = y + (yrow ? (long)b * N : 0);
        const float* qb = xnew + (qrow ? (long)b * P : 0);
        for (int p = 0; p < P; ++p) {
            const float q = qb[p];
            int lo = 0, hi = N;                   /* searchsorted, side='left': first index with x[idx] >= q */
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (xb[mid] < q) lo = mid + 1; else hi = mid;
            }
            int i = lo - 1;
            if (i < 0) i = 0;
            if (i > N - 2) i = N - 2;
            const float slope = (yb[i + 1] - yb[i]) / (eps + (xb[i + 1] - xb[i]));
            ynew[(long)b * P + p] = yb[i] + slope * (q - xb[i]);
            ind[(long)b * P + p] = i;
        }
    }
}

void cfn_ref_resize_index(int Tin, int L, int align_corners, int32_t* i0, int32_t* i1, float* lam) {
    for (int j = 0; j < L; ++j) {
        float src;
        if (align_corners) {
            const float scale = L > 1 ? (float)(Tin - 1) / (float)(L - 1) : 0.0f;
            src = scale * (float)j;
        } else {
            const float scale = (float)Tin / (float)L;
            src = fmaf(scale, (float)j + 0.5f, -0.5f);
            if (src < 0.0f) src = 0.0f;
        }
        const int a = (int)src;
        i0[j] = a;
        i1[j] = a + (a < Tin - 1 ? 1 : 0);
        lam[j] = src - (float)a;
    }
}
