// fo_scene.hip -- visibility ray-cast, occluded-cell grid, spawn sampling (placeholder until the kernels land).
#include "fo_ctx.hpp"
extern "C" void fo_scene_destroy_(fo_ctx *ctx) { (void)ctx; }
