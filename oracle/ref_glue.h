// ref_glue.h -- see ref_glue.cpp.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <vector>
#include "prt_oracle.h"
namespace prt { class Bvh; class Image; }
void refGlueRegister(const std::vector<prt::Bvh*>& bvhs, const std::vector<orc_mesh*>& omeshes);
prt::Image* refGlueMakeImage(uint32_t width, uint32_t height, float exposure);
const float* refGlueImagePixels(const prt::Image* image);
