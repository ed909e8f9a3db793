/*
 * agx_api.hip — error handling, device-memory helpers and stream timers of the C ABI (include/agx.h).
 */
#include "agx_internal.hpp"

#include <cstring>

namespace
{
	thread_local char g_last_error[1024] = "";
}

namespace agx
{
	void set_error(const char *fmt, ...)
	{
		va_list args;
		va_start(args, fmt);
		vsnprintf(g_last_error, sizeof(g_last_error), fmt, args);
		va_end(args);
	}
}

struct AgxTimer
{
		hipEvent_t start = nullptr;
		hipEvent_t stop = nullptr;
};

extern "C" {

const char* agx_last_error(void)
{
	return g_last_error;
}
int agx_version(void)
{
	return 1;
}
int agx_set_device(int device)
{
	AGX_HIP_CHECK(hipSetDevice(device));
	return AGX_OK;
}
int agx_device_count(int *count)
{
	AGX_REQUIRE(count != nullptr, AGX_ERR_INVALID, "agx_device_count: null argument");
	AGX_HIP_CHECK(hipGetDeviceCount(count));
	return AGX_OK;
}
int agx_device_cu_count(int *count)
{ // compute units of the current device (what a CU mask of agx_stream_create_with_cu_mask indexes)
	AGX_REQUIRE(count != nullptr, AGX_ERR_INVALID, "agx_device_cu_count: null argument");
	int device = 0;
	AGX_HIP_CHECK(hipGetDevice(&device));
	AGX_HIP_CHECK(hipDeviceGetAttribute(count, hipDeviceAttributeMultiprocessorCount, device));
	return AGX_OK;
}
int agx_malloc(void **d_ptr, size_t bytes)
{
	AGX_REQUIRE(d_ptr != nullptr, AGX_ERR_INVALID, "agx_malloc: null output pointer");
	AGX_HIP_CHECK(hipMalloc(d_ptr, bytes));
	return AGX_OK;
}
int agx_free(void *d_ptr)
{
	AGX_HIP_CHECK(hipFree(d_ptr));
	return AGX_OK;
}
int agx_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes)
{
	AGX_HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
	return AGX_OK;
}
int agx_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes)
{
	AGX_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
	return AGX_OK;
}
int agx_memset(void *d_ptr, int value, size_t bytes)
{
	AGX_HIP_CHECK(hipMemset(d_ptr, value, bytes));
	return AGX_OK;
}
int agx_device_synchronize(void)
{
	AGX_HIP_CHECK(hipDeviceSynchronize());
	return AGX_OK;
}

int agx_timer_create(AgxTimer **out)
{
	AGX_REQUIRE(out != nullptr, AGX_ERR_INVALID, "agx_timer_create: null output pointer");
	AgxTimer *t = new AgxTimer();
	hipError_t e = hipEventCreate(&t->start);
	if (e == hipSuccess)
		e = hipEventCreate(&t->stop);
	if (e != hipSuccess)
	{
		agx::set_error("hipEventCreate failed: %s", hipGetErrorString(e));
		delete t;
		return AGX_ERR_HIP;
	}
	*out = t;
	return AGX_OK;
}
int agx_timer_start(AgxTimer *t, void *stream)
{
	AGX_REQUIRE(t != nullptr, AGX_ERR_INVALID, "agx_timer_start: null timer");
	AGX_HIP_CHECK(hipEventRecord(t->start, static_cast<hipStream_t>(stream)));
	return AGX_OK;
}
int agx_timer_stop(AgxTimer *t, void *stream)
{
	AGX_REQUIRE(t != nullptr, AGX_ERR_INVALID, "agx_timer_stop: null timer");
	AGX_HIP_CHECK(hipEventRecord(t->stop, static_cast<hipStream_t>(stream)));
	return AGX_OK;
}
int agx_timer_elapsed_ms(AgxTimer *t, float *ms)
{
	AGX_REQUIRE(t != nullptr && ms != nullptr, AGX_ERR_INVALID, "agx_timer_elapsed_ms: null argument");
	AGX_HIP_CHECK(hipEventSynchronize(t->stop));
	AGX_HIP_CHECK(hipEventElapsedTime(ms, t->start, t->stop));
	return AGX_OK;
}
int agx_timer_poll_ms(AgxTimer *t, float *ms, int *ready)
{
	AGX_REQUIRE(t != nullptr && ms != nullptr && ready != nullptr, AGX_ERR_INVALID, "agx_timer_poll_ms: null argument");
	const hipError_t st = hipEventQuery(t->stop);
	if (st == hipErrorNotReady)
	{
		*ready = 0;
		return AGX_OK;
	}
	AGX_HIP_CHECK(st);
	AGX_HIP_CHECK(hipEventElapsedTime(ms, t->start, t->stop));
	*ready = 1;
	return AGX_OK;
}
int agx_timer_destroy(AgxTimer *t)
{
	if (t == nullptr)
		return AGX_OK;
	(void) hipEventDestroy(t->start);
	(void) hipEventDestroy(t->stop);
	delete t;
	return AGX_OK;
}

} /* extern "C" */
