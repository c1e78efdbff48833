// Opt-in kernel timing brackets (hipEvent pairs on the launch stream).
#pragma once
#include <hip/hip_runtime.h>
void fvta_prof_begin(int id, hipStream_t s);
void fvta_prof_end(int id, int launches, hipStream_t s);
