// <hip/hip_fp16.h> for the full host build: the half types live in the runtime stub.
#pragma once
#include "hip_runtime.h"
