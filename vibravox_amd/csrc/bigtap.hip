// bigtap.hip -- tap4_kernel (tap4_kernel.h): single-piece (EBEN_MATH_BF16) instantiations and the launch dispatch.
#include "tap4_kernel.h"

namespace eben {

int tap4_launch_x3(const Tap3Plan& p, const Tap3Args& a, hipStream_t st);   // bigtap_x3.hip

int tap4_launch(const Tap3Plan& p, const Tap3Args& a, hipStream_t st) {
  if (a.in_mode || !a.xh) return fail(EBEN_EUNSUPPORTED, "tap4: bundle-layout launches only");
  if (a.pr_S > 0 && (a.pr_S != 4 || a.pr_order != 1 || a.bias)) return fail(EBEN_EUNSUPPORTED, "tap4: phases as rows at stride 4 in bundle-major row order only");
  if (!a.id_fast || a.pg_n < a.nph) return fail(EBEN_EUNSUPPORTED, "tap4: grid beyond the tile decomposition's 32-bit arithmetic");
  if (a.bias && (reinterpret_cast<unsigned long long>(a.bias) & 15ull)) return fail(EBEN_EINVAL, "tap4: the bias vector must be 16-byte aligned");
  const int key = ((p.WM * 10 + p.WN) * 10 + p.TM) * 10 + p.TN;
  if (p.npw == 1 && p.KSC == 4 && p.RING == 3) {
    switch (key) {
      case 2242: return launch4<2, 2, 4, 2, 1, 4, 3, EBEN_T4_HB>(p, a, st);
      case 2232: return launch4<2, 2, 3, 2, 1, 4, 3, EBEN_T4_HB>(p, a, st);
      case 2222: return launch4<2, 2, 2, 2, 1, 4, 3, EBEN_T4_HB>(p, a, st);
      default: break;
    }
  } else if (p.npw == 2 && p.KSC == 2 && p.RING == 3) {
    return tap4_launch_x3(p, a, st);
  }
  return fail(EBEN_EUNSUPPORTED, "tap4: no instantiation for wave grid %d x %d, wave tile %d x %d, %d pieces, %d k-steps per chunk, ring %d", p.WM, p.WN,
              p.TM, p.TN, p.npw, p.KSC, p.RING);
}

}  // namespace eben

