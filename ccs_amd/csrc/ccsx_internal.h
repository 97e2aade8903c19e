// ccsx_internal.h — shared between the host and device translation units of libccsx.so
#pragma once
#include <cstdint>
#include <string>

void ccsx_set_error(const std::string &s);

// capacity of the draft / consensus of a ZMW whose longest subread has maxL bases
static inline int64_t ccsx_draft_cap(int64_t maxL) { return maxL + maxL / 4 + 64; }
// POA vertex capacity for the same ZMW
static inline int64_t ccsx_vertex_cap(int64_t maxL) { return (5 * maxL) / 2 + 256; }
