/* conv_ref.c -- TEST INFRASTRUCTURE (part of the CPU oracle, never linked into the product).
 *
 * Plain-C, double-precision restatement of the two ATen primitives every EBEN layer reduces to
 * (reference call sites: vibravox/torch_modules/dnn/eben_generator.py:112-166,241-249,272-280,
 * 295-312; eben_discriminator.py:66-157; melgan_discriminator.py:89-156; dsp/pqmf.py:194-213):
 *   conv1d (zero or reflect padding, stride, dilation, groups) and conv_transpose1d.
 * It exists so that the torch-based oracle (oracle/eben_oracle.py) is itself cross-checked by an
 * implementation that shares no code with PyTorch (tests/test_host.py).  Naive loops: small cases only.
 */
#include <stddef.h>

static long reflect_index(long q, long n) {
  if (q < 0) q = -q;
  if (q >= n) q = 2 * (n - 1) - q;
  return q;
}

/* y[b,co,t] = bias[co] + sum_{ci,k} w[co,ci,k] * xpad[b, g*Cg+ci, t*stride - pad_l + k*dil] */
void ref_conv1d(const double* x, const double* w, const double* bias, double* y, int B, int Cin, int Cout, int Lin, int Lout,
                int K, int stride, int dil, int groups, int pad_l, int reflect) {
  const int cg = Cin / groups, mg = Cout / groups;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      const int g = co / mg;
      for (int t = 0; t < Lout; ++t) {
        double acc = bias ? bias[co] : 0.0;
        for (int ci = 0; ci < cg; ++ci)
          for (int k = 0; k < K; ++k) {
            long q = (long)t * stride - pad_l + (long)k * dil;
            if (reflect) q = reflect_index(q, Lin);
            if (q < 0 || q >= Lin) continue;
            acc += w[((size_t)co * cg + ci) * K + k] * x[((size_t)b * Cin + g * cg + ci) * Lin + q];
          }
        y[((size_t)b * Cout + co) * Lout + t] = acc;
      }
    }
}

/* y[b, g*Og+o, u] = sum_{ci in group g, k, t : t*stride - pad + k*dil == u} w[ci,o,k] * x[b,ci,t]
 * weight layout (Cin, Cout/groups, K) as nn.ConvTranspose1d */
void ref_conv_transpose1d(const double* x, const double* w, double* y, int B, int Cin, int Cout, int Lin, int Lout, int K,
                          int stride, int dil, int groups, int pad) {
  const int cg = Cin / groups, og = Cout / groups;
  for (size_t i = 0; i < (size_t)B * Cout * Lout; ++i) y[i] = 0.0;
  for (int b = 0; b < B; ++b)
    for (int ci = 0; ci < Cin; ++ci) {
      const int g = ci / cg;
      for (int t = 0; t < Lin; ++t) {
        const double xv = x[((size_t)b * Cin + ci) * Lin + t];
        for (int o = 0; o < og; ++o)
          for (int k = 0; k < K; ++k) {
            const long u = (long)t * stride - pad + (long)k * dil;
            if (u < 0 || u >= Lout) continue;
            y[((size_t)b * Cout + g * og + o) * Lout + u] += w[((size_t)ci * og + o) * K + k] * xv;
          }
      }
    }
}
