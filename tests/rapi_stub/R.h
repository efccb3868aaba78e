/* tests/rapi_stub/R.h -- see Rinternals.h in this directory (test stand-in for R's header of the same name). */
#ifndef RAPI_STUB_R_H
#define RAPI_STUB_R_H
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#endif
