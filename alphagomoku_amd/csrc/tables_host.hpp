/*
 * tables_host.hpp — lookup tables built on the host and uploaded once per engine (see tables_host.cpp).
 */
#ifndef AGX_TABLES_HOST_HPP_
#define AGX_TABLES_HOST_HPP_

#include <cstdint>
#include <vector>

#include "../../include/agx.h"

namespace agx
{
	struct HostTables
	{
			int rules = 0;
			std::vector<uint8_t> pattern;          // [1 << 20] cross | circle << 4
			std::vector<uint8_t> half_open_three;  // [1 << 20] bit0 cross, bit1 circle
			std::vector<uint8_t> threat;           // [4096][2]
			std::vector<uint16_t> defense;         // [15][256][2]: rows 0-4 five, 5-8 open four, 9-14 double four; [..][0] cross defends
	};
	void build_host_tables(int rules, HostTables &out);
}

#endif
