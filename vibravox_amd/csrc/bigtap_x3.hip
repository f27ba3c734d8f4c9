// bigtap_x3.hip -- tap4_kernel (tap4_kernel.h): the hi + lo operand instantiations (EBEN_MATH_BF16X3: three MFMAs per product; the
// PQMF-band forwards).  A translation unit of its own: each instantiation takes hipcc ~15 s.
#include "tap4_kernel.h"

namespace eben {

int tap4_launch_x3(const Tap3Plan& p, const Tap3Args& a, hipStream_t st) {
  const int key = ((p.WM * 10 + p.WN) * 10 + p.TM) * 10 + p.TN;
  switch (key) {
    case 2242: return launch4<2, 2, 4, 2, 2, 2, 3, 1>(p, a, st);
    case 2232: return launch4<2, 2, 3, 2, 2, 2, 3, 1>(p, a, st);
    case 2222: return launch4<2, 2, 2, 2, 2, 2, 3, 1>(p, a, st);
    default: break;
  }
  return fail(EBEN_EUNSUPPORTED, "tap4: no hi + lo instantiation for wave grid %d x %d, wave tile %d x %d", p.WM, p.WN, p.TM, p.TN);
}

}  // namespace eben
