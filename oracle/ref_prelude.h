// ref_prelude.h -- pre-included when compiling the reference sources (see Makefile): headers nvcc pre-includes,
// and std::powf which MSVC declares but libstdc++ 11 does not (RayGen.cuh:59, Texture.cu:56).
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstdlib>
namespace std { using ::powf; }
