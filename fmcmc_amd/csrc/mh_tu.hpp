// mh_tu.hpp -- what every translation unit of libfmcmc_amd.so starts with.  The device code is split over several .hip files so
// that they compile in parallel (fmcmc_amd/build.py): mh_engine.hip holds the C-ABI, validation and kernel selection, every
// k_*.hip instantiates one kernel family and hands its kernels out by (run-time) shape through the look-ups of mh_kernels.hpp.
// All device helpers live in anonymous namespaces: each translation unit has its own copy, nothing device-side is linked.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <type_traits>

#include "../../include/fmcmc_amd.h"
#include "../../include/fmh_detmath.h"
#include "../../include/fmh_philox.h"

#include "mh_common.hpp"
#include "mh_rng.hpp"
#include "mh_kernels.hpp"
