// Mesher device structures + launchers (filled in by mesh_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

struct MeshDev { int placeholder; };
struct MeshHost { int placeholder; };
