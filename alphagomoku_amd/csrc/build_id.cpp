/*
 * build_id.cpp — the identity of this build of libagx.so: the hash of the sources it was compiled from
 * (alphagomoku_amd/build.py:source_hash writes build_id.inc before every link that follows a source change).
 * bench.py and the test suite refuse a library whose hash differs from the sources beside it.
 */
extern "C" const char* agx_build_hash(void)
{
	return
#include "build_id.inc"
	;
}
