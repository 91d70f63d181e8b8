"""The one documented float32 exponential used for the boost factor.

Reference: `regularizations.py:16` computes `np.exp(float32 array)`, whose SIMD
implementation is not correctly rounded (measured in the build container: 39 % of inputs
differ from the correctly rounded result, by up to 2 ulp) and depends on NumPy's CPU
dispatch, so it cannot be matched bit-for-bit by any other machine.  Policy (SURVEY.md
§8c): evaluate ONE fixed sequence of IEEE-754 double operations on both CPU and GPU and
round the double once to float32.  No fused multiply-add, no library `exp`:

    x   = (double) x32
    n   = rint(x * LOG2E)                      (round-half-even)
    r   = (x - n * LN2_HI) - n * LN2_LO        (Cody-Waite, n * LN2_HI is exact)
    p   = sum_{j=0..14} r**j / j!              (Horner, separate multiply and add)
    y32 = (float) ldexp(p, n)

|r| <= 0.347, so the Taylor tail is < 4e-18 and p is within ~1 ulp(double) of exp(r); the
single final rounding makes y32 the correctly rounded float32 exponential except when the
true value lies within ~2**-52 (relative) of a float32 rounding boundary.
Domain: x32 in [-87, 88] (float32-normal results).  The boost argument is
-(intensity / density) * duty with duty in [0, 1], i.e. [-15, 0] with default settings.

bithtm_amd/csrc/htm_fexp.h is the device twin (compiled with -ffp-contract=off).
"""

import numpy as np

LOG2E = float.fromhex("0x1.71547652b82fep+0")
LN2_HI = float.fromhex("0x1.62e42fee00000p-1")
LN2_LO = float.fromhex("0x1.a39ef35793c76p-33")
TAYLOR = tuple(float.fromhex(h) for h in (
    "0x1.0000000000000p+0", "0x1.0000000000000p+0", "0x1.0000000000000p-1",
    "0x1.5555555555555p-3", "0x1.5555555555555p-5", "0x1.1111111111111p-7",
    "0x1.6c16c16c16c17p-10", "0x1.a01a01a01a01ap-13", "0x1.a01a01a01a01ap-16",
    "0x1.71de3a556c734p-19", "0x1.27e4fb7789f5cp-22", "0x1.ae64567f544e4p-26",
    "0x1.1eed8eff8d898p-29", "0x1.6124613a86d09p-33", "0x1.93974a8c07c9dp-37",
))


def exp_f32(x32):
    """float32 ndarray -> float32 ndarray, see module docstring."""
    x = np.asarray(x32, dtype=np.float32).astype(np.float64)
    n = np.rint(x * LOG2E)
    r = (x - n * LN2_HI) - n * LN2_LO
    p = np.full_like(r, TAYLOR[14])
    for j in range(13, -1, -1):
        p = p * r + TAYLOR[j]
    y = np.ldexp(p, n.astype(np.int32))
    return y.astype(np.float32)
