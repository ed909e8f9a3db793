/*
 * agx_internal.hpp — shared helpers of the HIP/C++ implementation behind include/agx.h.
 */
#ifndef AGX_INTERNAL_HPP_
#define AGX_INTERNAL_HPP_

#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>
#include <cstdarg>

#include "../../include/agx.h"

namespace agx
{
	void set_error(const char *fmt, ...);

	struct HipError
	{
			hipError_t code;
	};
}

#define AGX_HIP_CHECK(expr)                                                                         \
	do {                                                                                            \
		hipError_t _e = (expr);                                                                     \
		if (_e != hipSuccess) {                                                                     \
			agx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
			return AGX_ERR_HIP;                                                                     \
		}                                                                                           \
	} while (0)

#define AGX_REQUIRE(cond, code, ...)                                                                \
	do {                                                                                            \
		if (!(cond)) {                                                                              \
			agx::set_error(__VA_ARGS__);                                                            \
			return (code);                                                                          \
		}                                                                                           \
	} while (0)

#endif /* AGX_INTERNAL_HPP_ */
