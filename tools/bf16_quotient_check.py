#!/usr/bin/env python3
"""Exhaustive check behind dis_kernel's bf16 normalisation: for bf16 x and a normal bf16 divisor n,
bf16(fl32(x * fl32(1/n))) == bf16(fl32(x / n)) for every mantissa pair.  The exact quotient of two 8-bit
significands is at least 1/(255*512) away (relative to [0.5,2)) from any 9-bit rounding midpoint, i.e. >= 128 fp32
ulp, while x*rcp(n) is within 2 fp32 ulp of it -- so the reciprocal product can never round differently.
Scaling by powers of two is exact, so mantissa pairs x a few exponent offsets (including results in the bf16
subnormal range) cover everything."""
import numpy as np


def bf16_rne(f32):
    u = f32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.astype(np.uint32).view(np.float32)


def main():
    mant = np.arange(128, 256, dtype=np.float64)
    worst = np.inf
    bad = 0
    for ex in (0, 1, -1, 7, -7, 60, -60, -120, -126, -130, -133):      # exponent of x relative to n
        for en in (0, 10, -10, 100, -26):                                  # exponent of n (normal range)
            x = (mant[:, None] * 2.0 ** (ex + en - 7)).astype(np.float32)  # bf16-representable (8-bit significands)
            n = (mant[None, :] * 2.0 ** (en - 7)).astype(np.float32)
            with np.errstate(under="ignore", over="ignore"):
                x, n = np.broadcast_arrays(x, n)
                ok = np.isfinite(x) & np.isfinite(n) & (n > 0)
                rn = (np.float32(1.0) / n).astype(np.float32)
                fast = bf16_rne((x * rn).astype(np.float32))
                ref = bf16_rne((x / n).astype(np.float32))
                bad += int((fast[ok] != ref[ok]).sum())
                # distance of the exact quotient to the nearest bf16 midpoint, in fp32 ulp of the quotient
                q = x.astype(np.float64) / n.astype(np.float64)
                nz = ok & (q > 2.0 ** -120)
                e = np.floor(np.log2(q[nz]))
                frac = q[nz] / 2.0 ** e * 128.0          # in bf16 ulps, [128, 256)
                d = np.abs(frac - np.floor(frac) - 0.5)  # distance to the midpoint, in bf16 ulps
                exact = (frac == np.floor(frac))
                if (~exact).any():
                    worst = min(worst, float(d[~exact].min()) * 65536.0)
    # subnormal x significands (1..127) too
    for ex in (-133,):
        x = (np.arange(1, 128, dtype=np.float64)[:, None] * 2.0 ** ex).astype(np.float32)
        for en in (0, -3, 3):
            n = (mant[None, :] * 2.0 ** (en - 7)).astype(np.float32)
            xb, nb = np.broadcast_arrays(x, n)
            rn = (np.float32(1.0) / nb).astype(np.float32)
            bad += int((bf16_rne((xb * rn).astype(np.float32)) != bf16_rne((xb / nb).astype(np.float32))).sum())
    print("mismatches:", bad, " closest non-exact quotient to a bf16 midpoint: %.1f fp32 ulp" % worst)
    return bad


if __name__ == "__main__":
    raise SystemExit(1 if main() else 0)
