"""Generates the tables of fmh_log1p_exp_nonpos (include/fmh_detmath.h) from 70-digit decimal arithmetic, as C initialiser
rows with hexadecimal floating literals (exact):
  (1) logarithm: for i = 0..127, invc_i = fl(1 / (1 + i/128)) and logc_i = -log(invc_i) as a double-double (hi, lo):
      u in [1 + i/128, 1 + (i+1)/128):  log(u) = logc_i + log1p(r),  r = u * invc_i - 1 in [~0, 2^-7];
  (2) exponential: for j = 0..127, 2^(j/128) as a double-double (hi, lo);
  (3) ln2/128 split into a 32-bit head (k * head is exact for |k| < 2^21) and a tail, and 128/ln2.
Usage: python tools/gen_softplus_table.py [log|exp|consts]"""
import sys
from decimal import Decimal, getcontext
getcontext().prec = 70
which = sys.argv[1] if len(sys.argv) > 1 else "log"
LN2 = Decimal(2).ln()
if which == "log":
    for i in range(128):
        c = 1.0 + i / 128.0
        invc = 1.0 / c                       # correctly rounded double
        d = Decimal(invc)                    # exact value of the double
        logc = -d.ln()
        hi = float(logc)                     # round to nearest double
        lo = float(logc - Decimal(hi))
        if i == 0:
            invc, hi, lo = 1.0, 0.0, 0.0
        print("  %s, %s, %s," % (invc.hex(), hi.hex(), lo.hex()))
elif which == "exp":
    for j in range(128):
        v = (LN2 * j / 128).exp()
        hi = float(v)
        lo = float(v - Decimal(hi))
        print("  %s, %s," % (hi.hex(), lo.hex()))
else:
    import struct, math
    h = float(LN2 / 128)
    bits = struct.unpack("<Q", struct.pack("<d", h))[0] & ~((1 << 21) - 1)     # keep 32 significant bits
    head = struct.unpack("<d", struct.pack("<Q", bits))[0]
    tail = float(LN2 / 128 - Decimal(head))
    print("LN2_N_HI", head.hex(), "LN2_N_LO", tail.hex(), "N_INV_LN2", float(Decimal(128) / LN2).hex())
    for n in range(2, 6):
        print("E%d" % n, (1.0 / math.factorial(n)).hex())
