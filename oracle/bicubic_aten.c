/* ORACLE - test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
 * never by the product).
 *
 * Bit-exact CPU restatement of the float64 bicubic resize the reference reaches through
 *   network/computations.py:308-311  resize(): F.interpolate(depth_map.double(), size, mode='bicubic', align_corners=False)
 * The arithmetic itself lives in a third-party dependency that is not under /root/reference: torch (unpinned in the
 * reference's requirements.txt; the build container and the GPU box carry torch 2.10.0), ATen
 *   aten/src/ATen/native/UpSample.h            area_pixel_compute_source_index, guard_index_and_lambda,
 *                                              get_cubic_upsample_coefficients, cubic_convolution1/2 (A = -0.75)
 *   aten/src/ATen/native/cpu/UpSampleKernel.cpp cpu_upsample_generic / Interpolate<2, double>::eval (H outer, W inner)
 * Published algorithm: Keys cubic convolution, 4 taps per axis, border indices clamped, no antialiasing.
 * What the published formulae leave open is the ROUNDING sequence (which multiply-adds the library's compiler fused).
 * It was pinned by running the library in the build container against every candidate sequence (exact-rational
 * emulation of each fused / unfused step): exactly one reproduces it bit for bit on all 3 318 + 13 fixture outputs:
 *   real = fma(scale, i + 0.5, -0.5)               scale = in / out
 *   idx  = min((int64) floorf((float) real), in - 1)    (ATen floors in FLOAT)
 *   t    = min(max(real - idx, 0), 1)
 *   c2(x) = fma(fma(A, x, -5A), x, 8A) * x - 4A     outer taps, x = t + 1 and (1 - t) + 1
 *   c1(x) = fma(A + 2, x, -(A + 3)) * x * x + 1     inner taps, x = t and 1 - t
 *   dot4(v, w) = fma(v3, w3, fma(v2, w2, fma(v0, w0, v1 * w1)))
 * Compile with -ffp-contract=off: every fused step is an explicit fma(), everything else rounds separately.
 * Pinned by tests/test_oracle_ops.py::test_resize against tests/golden/op_goldens.npz (outputs of the reference's own
 * cp.resize) with assert_array_equal.
 */
#include <math.h>
#include <stdint.h>

static void coeffs(double t, double c[4]) {
  const double A = -0.75;
  const double x2 = 1.0 - t;
  const double xa = t + 1.0, xb = x2 + 1.0;
  c[0] = fma(fma(A, xa, -5.0 * A), xa, 8.0 * A) * xa - 4.0 * A;
  c[1] = fma(A + 2.0, t, -(A + 3.0)) * t * t + 1.0;
  c[2] = fma(A + 2.0, x2, -(A + 3.0)) * x2 * x2 + 1.0;
  c[3] = fma(fma(A, xb, -5.0 * A), xb, 8.0 * A) * xb - 4.0 * A;
}

static int64_t source_index(int64_t i, int64_t n_in, int64_t n_out, double* t) {
  const double scale = (double)n_in / (double)n_out;
  const double real = fma(scale, (double)i + 0.5, -0.5);
  int64_t idx = (int64_t)floorf((float)real);
  if (idx > n_in - 1) idx = n_in - 1;
  double lam = real - (double)idx;
  if (lam < 0.0) lam = 0.0;
  if (lam > 1.0) lam = 1.0;
  *t = lam;
  return idx;
}

static double dot4(const double v[4], const double w[4]) {
  double acc = v[1] * w[1];
  acc = fma(v[0], w[0], acc);
  acc = fma(v[2], w[2], acc);
  return fma(v[3], w[3], acc);
}

static int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* src (n, h, w) -> dst (n, oh, ow), float64, contiguous */
void rdm_oracle_resize_bicubic_f64(const double* src, double* dst, int64_t n, int64_t h, int64_t w, int64_t oh, int64_t ow) {
  for (int64_t b = 0; b < n; ++b) {
    const double* s = src + b * h * w;
    for (int64_t oy = 0; oy < oh; ++oy) {
      double ty, cy[4];
      const int64_t iy = source_index(oy, h, oh, &ty);
      coeffs(ty, cy);
      for (int64_t ox = 0; ox < ow; ++ox) {
        double tx, cx[4], rows[4];
        const int64_t ix = source_index(ox, w, ow, &tx);
        coeffs(tx, cx);
        for (int i = 0; i < 4; ++i) {
          const int64_t y = clampi(iy - 1 + i, 0, h - 1);
          double v[4];
          for (int j = 0; j < 4; ++j) v[j] = s[y * w + clampi(ix - 1 + j, 0, w - 1)];
          rows[i] = dot4(v, cx);
        }
        dst[(b * oh + oy) * ow + ox] = dot4(rows, cy);
      }
    }
  }
}
