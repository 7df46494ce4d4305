// include/prt_compat/material.h -- lets a caller written against the reference's headers (its main.cpp includes "material.h",
// /root/reference/src/main.cpp:9-16) compile unchanged against this library: the whole host surface is one header.
#pragma once
#include "../../prt_amd/csrc/host/prt.h"
