"""Generates the 128-entry table of fmh_log1p_exp_nonpos (include/fmh_detmath.h): for i = 0..127,
   invc_i = fl(1 / (1 + i/128)) and logc_i = -log(invc_i) as a double-double (hi, lo), from 60-digit decimal arithmetic.
   u in [1 + i/128, 1 + (i+1)/128):  log(u) = logc_i + log1p(r),  r = u * invc_i - 1 in [~0, 2^-7].
   Output: C initialiser rows with hexadecimal floating literals (exact)."""
from decimal import Decimal, getcontext
getcontext().prec = 70
rows = []
for i in range(128):
    c = 1.0 + i / 128.0
    invc = 1.0 / c                       # correctly rounded double
    d = Decimal(invc)                    # exact value of the double
    logc = -d.ln()
    hi = float(logc)                     # round to nearest double
    lo = float(logc - Decimal(hi))
    if i == 0:
        invc, hi, lo = 1.0, 0.0, 0.0
    rows.append("  %s, %s, %s," % (invc.hex(), hi.hex(), lo.hex()))
print("\n".join(rows))
