// <cuda.h> stand-in: everything lives in cuda_runtime.h.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "cuda_runtime.h"
