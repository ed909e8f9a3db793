/*
 * game_buffer.cpp — host side of the self-play record sink: finished games framed as the reference's
 * GameDataStorage format 201 and collected in a GameDataBuffer.
 *
 *   AgxGameBuffer            <- ag::GameDataBuffer (src/dataset/GameDataBuffer.cpp) + GeneratorManager::addToBuffer under its
 *                               mutex (src/selfplay/GeneratorManager.cpp:160-164): several generator threads (one per device)
 *                               hand their finished games to one buffer
 *   game bytes               <- GameDataStorage::serialize, case 201 (src/dataset/GameDataStorage.cpp:217-250): u32 sample count,
 *                               the samples (SearchDataStorage_v201::serialize — produced ON THE DEVICE by k_advance, see
 *                               sample_v201.hpp), u32 move count + u16 Move::toShort of every move of the game (opening included),
 *                               int outcome, int rows, int cols
 *   agx_game_buffer_save     <- GameDataBuffer::save (:97-113): one line of JSON {"format": 201, "config": GameConfig::toJson,
 *                               "offsets": [...]}, '\n', the concatenated game bytes.  The reference then compresses the whole
 *                               file with MinML's ZipWrapper, which is not in the reference tree (format unpinned): compress = 1
 *                               writes a zlib stream of the same content, compress = 0 the plain bytes.
 *   agx_sample_v201_unpack   <- SearchDataStorage_v201(const SerializedObject&, size_t&) + storeTo (SearchDataStorage.cpp:300-320,
 *                               375-409): what a consumer of the buffer (GameDataStorage::getSample) reads back
 */
#include "agx_internal.hpp"
#include "sample_v201.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace
{
	struct PendingSample
	{
			int move_number;
			std::vector<uint8_t> bytes;
	};
	struct PendingGame
	{
			std::vector<PendingSample> samples;
	};
	template<typename T>
	void put(std::vector<uint8_t> &out, T v)
	{
		const uint8_t *p = reinterpret_cast<const uint8_t*>(&v);
		out.insert(out.end(), p, p + sizeof(T));
	}
	const char* rules_name(int rules)
	{ // toString(GameRules) (src/game/rules.cpp)
		static const char *names[] = { "FREESTYLE", "STANDARD", "RENJU", "CARO5", "CARO6" };
		return names[rules];
	}
}

struct AgxGameBuffer
{
		int rules, rows, cols, draw_after;
		mutable std::mutex mutex; // GeneratorManager::buffer_mutex
		std::vector<std::vector<uint8_t>> games; // GameDataStorage::serialize bytes
		std::vector<int> outcomes, lengths, samples;
		// samples of games still being played, per producer (engine) and game (slot, index)
		std::map<const void*, std::map<std::pair<int, int>, PendingGame>> pending;
};

extern "C" {

int agx_game_buffer_create(int rules, int rows, int cols, int draw_after, AgxGameBuffer **out)
{
	AGX_REQUIRE(out != nullptr, AGX_ERR_INVALID, "agx_game_buffer_create: null argument");
	AGX_REQUIRE(rules >= 0 && rules <= AGX_CARO6 && rows > 0 && cols > 0, AGX_ERR_INVALID, "agx_game_buffer_create: invalid game configuration");
	AgxGameBuffer *b = new AgxGameBuffer();
	b->rules = rules;
	b->rows = rows;
	b->cols = cols;
	b->draw_after = (draw_after > 0) ? draw_after : rows * cols;
	*out = b;
	return AGX_OK;
}
int agx_game_buffer_destroy(AgxGameBuffer *b)
{
	delete b;
	return AGX_OK;
}
int agx_game_buffer_clear(AgxGameBuffer *b)
{
	AGX_REQUIRE(b != nullptr, AGX_ERR_INVALID, "agx_game_buffer_clear: null buffer");
	std::lock_guard<std::mutex> lock(b->mutex);
	b->games.clear();
	b->outcomes.clear();
	b->lengths.clear();
	b->samples.clear();
	return AGX_OK;
}

int agx_game_buffer_collect(AgxGameBuffer *b, AgxEngine *engine, int *games_added)
{
	AGX_REQUIRE(b != nullptr && engine != nullptr, AGX_ERR_INVALID, "agx_game_buffer_collect: null argument");
	AgxRecordCounts counts;
	int st = agx_engine_fetch_records(engine, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, &counts, 0);
	if (st != AGX_OK)
		return st;
	std::vector<AgxMoveRecord> records(std::max(counts.records, 1));
	std::vector<uint8_t> bytes(std::max(counts.sample_bytes, 4));
	std::vector<AgxGameEnd> ends(std::max(counts.game_ends, 1));
	// the device is idle between the two calls (one host thread drives an engine), so the counts cannot have grown
	st = agx_engine_fetch_records(engine, records.data(), static_cast<int>(records.size()), nullptr, 0, bytes.data(), static_cast<int>(bytes.size()), ends.data(),
			static_cast<int>(ends.size()), &counts, 1);
	if (st != AGX_OK)
		return st;
	AGX_REQUIRE(counts.records == 0 || counts.sample_bytes > 0, AGX_ERR_STATE,
			"agx_game_buffer_collect: the engine does not record format-201 samples (AgxEngineConfig.record_format bit 1)");

	std::lock_guard<std::mutex> lock(b->mutex); // GeneratorManager::addToBuffer
	std::map<std::pair<int, int>, PendingGame> &mine = b->pending[engine];
	int missing_samples = 0;
	for (int i = 0; i < counts.records; i++)
	{ // GameDataStorage::addSample (GameGenerator.cpp:170)
		const AgxMoveRecord &r = records[i];
		if (r.sample_offset < 0)
		{ // the sample did not fit into the device's sample pool: the game would be saved with fewer samples than moves — reported below
			missing_samples++;
			continue;
		}
		PendingSample s;
		s.move_number = r.move_number;
		s.bytes.assign(bytes.begin() + r.sample_offset, bytes.begin() + r.sample_offset + r.sample_bytes);
		mine[std::make_pair(r.game_slot, r.game_index)].samples.push_back(std::move(s));
	}
	int added = 0;
	for (int i = 0; i < counts.game_ends; i++)
	{ // setOutcome, addMoves(game.getMoves()), addToBuffer when the game holds samples (GameGenerator.cpp:104-111)
		const AgxGameEnd &g = ends[i];
		auto it = mine.find(std::make_pair(g.game_slot, g.game_index));
		if (it == mine.end())
			continue;
		PendingGame game = std::move(it->second);
		mine.erase(it);
		if (game.samples.empty())
			continue;
		std::stable_sort(game.samples.begin(), game.samples.end(), [](const PendingSample &x, const PendingSample &y) { return x.move_number < y.move_number; });
		std::vector<uint8_t> out;
		put<uint32_t>(out, static_cast<uint32_t>(game.samples.size()));
		for (const PendingSample &s : game.samples)
			out.insert(out.end(), s.bytes.begin(), s.bytes.end());
		put<uint32_t>(out, static_cast<uint32_t>(g.n_moves));
		for (int k = 0; k < g.n_moves; k++)
			put<uint16_t>(out, g.moves[k]);
		put<int>(out, g.outcome);
		put<int>(out, b->rows);
		put<int>(out, b->cols);
		b->games.push_back(std::move(out));
		b->outcomes.push_back(g.outcome);
		b->lengths.push_back(g.n_moves);
		b->samples.push_back(static_cast<int>(game.samples.size()));
		added++;
	}
	if (games_added != nullptr)
		*games_added = added;
	// games the engine stopped with an error never end: their samples must not stay pending for ever
	AgxEngineStats stats;
	if (agx_engine_stats(engine, &stats) == AGX_OK && stats.first_error != 0)
	{
		for (auto it = mine.begin(); it != mine.end();)
		{
			AgxGameInfo info;
			if (agx_engine_game_info(engine, it->first.first, &info, nullptr, nullptr, 0) == AGX_OK && info.error != 0)
				it = mine.erase(it);
			else
				++it;
		}
	}
	AGX_REQUIRE(missing_samples == 0, AGX_ERR_STATE,
			"agx_game_buffer_collect: %d move records lost their format-201 sample (sample pool full: raise AgxEngineConfig.record_sample_capacity or collect more often)",
			missing_samples);
	return AGX_OK;
}

int agx_game_buffer_stats(const AgxGameBuffer *b, AgxGameBufferStats *out)
{ // GameDataBuffer::getStats (GameDataBuffer.cpp:132-158)
	AGX_REQUIRE(b != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_game_buffer_stats: null argument");
	std::lock_guard<std::mutex> lock(b->mutex);
	std::memset(out, 0, sizeof(*out));
	out->games = static_cast<int>(b->games.size());
	for (size_t i = 0; i < b->games.size(); i++)
	{
		out->samples += b->samples[i];
		out->game_length += b->lengths[i];
		out->cross_win += (b->outcomes[i] == 2);
		out->draws += (b->outcomes[i] == 1);
		out->circle_win += (b->outcomes[i] == 3);
	}
	return AGX_OK;
}

int agx_game_buffer_game(const AgxGameBuffer *b, int index, uint8_t *h_bytes, size_t capacity, size_t *size)
{
	AGX_REQUIRE(b != nullptr && size != nullptr, AGX_ERR_INVALID, "agx_game_buffer_game: null argument");
	std::lock_guard<std::mutex> lock(b->mutex);
	AGX_REQUIRE(index >= 0 && index < static_cast<int>(b->games.size()), AGX_ERR_INVALID, "agx_game_buffer_game: game %d out of range", index);
	*size = b->games[index].size();
	if (h_bytes == nullptr)
		return AGX_OK;
	AGX_REQUIRE(capacity >= *size, AGX_ERR_INVALID, "agx_game_buffer_game: %zu bytes do not fit into %zu", *size, capacity);
	std::memcpy(h_bytes, b->games[index].data(), *size);
	return AGX_OK;
}

int agx_game_buffer_save(const AgxGameBuffer *b, const char *path, int compress)
{
	AGX_REQUIRE(b != nullptr && path != nullptr, AGX_ERR_INVALID, "agx_game_buffer_save: null argument");
	std::lock_guard<std::mutex> lock(b->mutex);
	std::string json = "{\"format\": 201, \"config\": {\"rules\": \"" + std::string(rules_name(b->rules)) + "\", \"rows\": " + std::to_string(b->rows) + ", \"cols\": "
			+ std::to_string(b->cols) + ", \"draw_after\": " + std::to_string(b->draw_after) + "}, \"offsets\": [";
	size_t offset = 0;
	for (size_t i = 0; i < b->games.size(); i++)
	{
		json += (i ? ", " : "") + std::to_string(offset);
		offset += b->games[i].size();
	}
	json += "]}";
	std::vector<char> to_save(json.begin(), json.end());
	to_save.push_back('\n');
	for (const std::vector<uint8_t> &g : b->games)
		to_save.insert(to_save.end(), g.begin(), g.end());
	if (compress)
	{
		uLongf bound = compressBound(static_cast<uLong>(to_save.size()));
		std::vector<char> packed(bound);
		const int z = compress2(reinterpret_cast<Bytef*>(packed.data()), &bound, reinterpret_cast<const Bytef*>(to_save.data()), static_cast<uLong>(to_save.size()),
				Z_DEFAULT_COMPRESSION);
		AGX_REQUIRE(z == Z_OK, AGX_ERR_STATE, "agx_game_buffer_save: zlib failed with %d", z);
		packed.resize(bound);
		to_save.swap(packed);
	}
	std::ofstream stream(path, std::ofstream::out | std::ofstream::binary);
	AGX_REQUIRE(stream.good(), AGX_ERR_STATE, "agx_game_buffer_save: cannot open '%s'", path);
	stream.write(to_save.data(), static_cast<std::streamsize>(to_save.size()));
	AGX_REQUIRE(stream.good(), AGX_ERR_STATE, "agx_game_buffer_save: writing '%s' failed", path);
	return AGX_OK;
}

/* GameDataBuffer::load (GameDataBuffer.cpp:115-131) for files written by agx_game_buffer_save (zlib stream or plain): the games are APPENDED.
 * Outcome, length and sample count of a game are read back from its GameDataStorage bytes (format 201: u32 samples, the samples — each
 * 16 + 6 * count bytes —, u32 moves, u16 per move, int outcome, int rows, int cols). */
int agx_game_buffer_load(AgxGameBuffer *b, const char *path)
{
	AGX_REQUIRE(b != nullptr && path != nullptr, AGX_ERR_INVALID, "agx_game_buffer_load: null argument");
	std::ifstream stream(path, std::ifstream::in | std::ifstream::binary);
	AGX_REQUIRE(stream.good(), AGX_ERR_STATE, "agx_game_buffer_load: cannot open '%s'", path);
	std::vector<char> raw((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
	AGX_REQUIRE(!raw.empty(), AGX_ERR_INVALID, "agx_game_buffer_load: '%s' is empty", path);
	if (raw[0] != '{')
	{ // a zlib stream: inflate into a growing buffer
		std::vector<char> plain(std::max<size_t>(raw.size() * 4, 1 << 16));
		while (true)
		{
			uLongf size = static_cast<uLongf>(plain.size());
			const int z = uncompress(reinterpret_cast<Bytef*>(plain.data()), &size, reinterpret_cast<const Bytef*>(raw.data()), static_cast<uLong>(raw.size()));
			if (z == Z_BUF_ERROR)
			{
				plain.resize(plain.size() * 2);
				continue;
			}
			AGX_REQUIRE(z == Z_OK, AGX_ERR_INVALID, "agx_game_buffer_load: '%s' is neither a JSON header nor a zlib stream (zlib %d)", path, z);
			plain.resize(size);
			break;
		}
		raw.swap(plain);
	}
	const auto newline = std::find(raw.begin(), raw.end(), '\n');
	AGX_REQUIRE(newline != raw.end(), AGX_ERR_INVALID, "agx_game_buffer_load: '%s' has no header line", path);
	const std::string header(raw.begin(), newline);
	AGX_REQUIRE(header.find("\"format\": 201") != std::string::npos, AGX_ERR_UNSUPPORTED, "agx_game_buffer_load: only dataset format 201 is read");
	auto number_after = [&](const std::string &key, long long &out)
	{
		const size_t at = header.find(key);
		if (at == std::string::npos)
			return false;
		out = std::atoll(header.c_str() + at + key.size());
		return true;
	};
	long long rows = 0, cols = 0;
	AGX_REQUIRE(number_after("\"rows\": ", rows) && number_after("\"cols\": ", cols) && rows == b->rows && cols == b->cols, AGX_ERR_INVALID,
			"agx_game_buffer_load: '%s' holds %lldx%lld games, the buffer %dx%d", path, rows, cols, b->rows, b->cols);
	AGX_REQUIRE(header.find(std::string("\"rules\": \"") + rules_name(b->rules) + "\"") != std::string::npos, AGX_ERR_INVALID,
			"agx_game_buffer_load: '%s' was saved for other rules than %s", path, rules_name(b->rules));
	std::vector<size_t> offsets;
	{
		const size_t at = header.find("\"offsets\": [");
		AGX_REQUIRE(at != std::string::npos, AGX_ERR_INVALID, "agx_game_buffer_load: no offsets in the header");
		const char *p = header.c_str() + at + 12;
		while (*p != ']' && *p != 0)
		{
			char *end = nullptr;
			const unsigned long long v = std::strtoull(p, &end, 10);
			if (end == p)
				break;
			offsets.push_back(static_cast<size_t>(v));
			p = end;
			while (*p == ',' || *p == ' ')
				p++;
		}
	}
	const uint8_t *blob = reinterpret_cast<const uint8_t*>(&*newline) + 1;
	const size_t blob_size = static_cast<size_t>(raw.end() - newline) - 1;
	std::vector<std::vector<uint8_t>> games;
	std::vector<int> outcomes, lengths, samples;
	for (size_t i = 0; i < offsets.size(); i++)
	{
		const size_t begin = offsets[i], end = (i + 1 < offsets.size()) ? offsets[i + 1] : blob_size;
		// (subtraction forms: an offset near SIZE_MAX must not wrap the test; offsets must ascend)
		AGX_REQUIRE(begin <= blob_size && end <= blob_size && begin <= end && end - begin >= 20, AGX_ERR_INVALID, "agx_game_buffer_load: game %zu lies outside the file", i);
		const uint8_t *g = blob + begin;
		uint32_t n_samples = 0, n_moves = 0;
		std::memcpy(&n_samples, g, 4);
		size_t at = 4;
		for (uint32_t k = 0; k < n_samples; k++)
		{
			AGX_REQUIRE(at <= end - begin && end - begin - at >= static_cast<size_t>(agx::v201::HEADER_BYTES), AGX_ERR_INVALID, "agx_game_buffer_load: game %zu is truncated", i);
			uint32_t count = 0;
			std::memcpy(&count, g + at + 12, 4);
			at += agx::v201::HEADER_BYTES + static_cast<size_t>(agx::v201::ENTRY_BYTES) * count;
		}
		AGX_REQUIRE(at <= end - begin && end - begin - at >= 4, AGX_ERR_INVALID, "agx_game_buffer_load: game %zu is truncated", i);
		std::memcpy(&n_moves, g + at, 4);
		at += 4 + 2 * static_cast<size_t>(n_moves);
		AGX_REQUIRE(at <= end - begin && end - begin - at == 12, AGX_ERR_INVALID, "agx_game_buffer_load: game %zu has %zu bytes where its layout needs %zu", i, end - begin, at + 12);
		int outcome = 0, g_rows = 0, g_cols = 0;
		std::memcpy(&outcome, g + at, 4);
		std::memcpy(&g_rows, g + at + 4, 4);
		std::memcpy(&g_cols, g + at + 8, 4);
		AGX_REQUIRE(g_rows == b->rows && g_cols == b->cols, AGX_ERR_INVALID, "agx_game_buffer_load: game %zu is %dx%d, the buffer %dx%d", i, g_rows, g_cols, b->rows, b->cols);
		games.emplace_back(g, g + (end - begin));
		outcomes.push_back(outcome);
		lengths.push_back(static_cast<int>(n_moves));
		samples.push_back(static_cast<int>(n_samples));
	}
	std::lock_guard<std::mutex> lock(b->mutex);
	for (size_t i = 0; i < games.size(); i++)
	{
		b->games.push_back(std::move(games[i]));
		b->outcomes.push_back(outcomes[i]);
		b->lengths.push_back(lengths[i]);
		b->samples.push_back(samples[i]);
	}
	return AGX_OK;
}

/* The samples a game in flight has collected so far (GameDataStorage::serialize of GameGenerator::save, GameGenerator.cpp:127), as
 * { i32 move number, u32 bytes, the format-201 sample } records; they are REMOVED from the buffer's pending list: the engine that produced
 * them is about to be destroyed.  h_bytes NULL: only the size (nothing is removed). */
int agx_game_buffer_take_pending(AgxGameBuffer *b, const AgxEngine *engine, int game_slot, int game_index, uint8_t *h_bytes, size_t capacity, size_t *size)
{
	AGX_REQUIRE(b != nullptr && size != nullptr, AGX_ERR_INVALID, "agx_game_buffer_take_pending: null argument");
	std::lock_guard<std::mutex> lock(b->mutex);
	*size = 0;
	auto mine = b->pending.find(engine);
	if (mine == b->pending.end())
		return AGX_OK;
	auto it = mine->second.find(std::make_pair(game_slot, game_index));
	if (it == mine->second.end())
		return AGX_OK;
	std::vector<uint8_t> out;
	for (const PendingSample &s : it->second.samples)
	{
		put<int32_t>(out, s.move_number);
		put<uint32_t>(out, static_cast<uint32_t>(s.bytes.size()));
		out.insert(out.end(), s.bytes.begin(), s.bytes.end());
	}
	*size = out.size();
	if (h_bytes == nullptr)
		return AGX_OK;
	AGX_REQUIRE(capacity >= out.size(), AGX_ERR_INVALID, "agx_game_buffer_take_pending: %zu bytes do not fit into %zu", out.size(), capacity);
	std::memcpy(h_bytes, out.data(), out.size());
	mine->second.erase(it);
	if (mine->second.empty())
		b->pending.erase(mine);
	return AGX_OK;
}
/* ... and handed back for the game that continues it in another engine (GameGenerator::load, GameGenerator.cpp:131-136) */
int agx_game_buffer_restore_pending(AgxGameBuffer *b, const AgxEngine *engine, int game_slot, int game_index, const uint8_t *h_bytes, size_t size)
{
	AGX_REQUIRE(b != nullptr && engine != nullptr && (h_bytes != nullptr || size == 0), AGX_ERR_INVALID, "agx_game_buffer_restore_pending: null argument");
	PendingGame game;
	size_t at = 0;
	while (at < size)
	{
		AGX_REQUIRE(at + 8 <= size, AGX_ERR_INVALID, "agx_game_buffer_restore_pending: truncated record");
		int32_t move_number;
		uint32_t bytes;
		std::memcpy(&move_number, h_bytes + at, 4);
		std::memcpy(&bytes, h_bytes + at + 4, 4);
		at += 8;
		AGX_REQUIRE(bytes >= static_cast<uint32_t>(agx::v201::HEADER_BYTES) && at + bytes <= size, AGX_ERR_INVALID, "agx_game_buffer_restore_pending: truncated sample");
		PendingSample s;
		s.move_number = move_number;
		s.bytes.assign(h_bytes + at, h_bytes + at + bytes);
		game.samples.push_back(std::move(s));
		at += bytes;
	}
	std::lock_guard<std::mutex> lock(b->mutex);
	PendingGame &slot = b->pending[engine][std::make_pair(game_slot, game_index)];
	slot.samples.insert(slot.samples.begin(), game.samples.begin(), game.samples.end());
	return AGX_OK;
}
/* forgets what an engine left pending (call before agx_engine_destroy: another engine may be created at the same address) */
int agx_game_buffer_forget_engine(AgxGameBuffer *b, const AgxEngine *engine)
{
	AGX_REQUIRE(b != nullptr, AGX_ERR_INVALID, "agx_game_buffer_forget_engine: null buffer");
	std::lock_guard<std::mutex> lock(b->mutex);
	b->pending.erase(engine);
	return AGX_OK;
}

int agx_sample_v201_unpack(const uint8_t *h_bytes, size_t size, int rows, int cols, int32_t *visits, float *prior, float *value, uint16_t *score, int *header,
		float *minimax_value, size_t *consumed)
{
	using namespace agx::v201;
	AGX_REQUIRE(h_bytes != nullptr && visits != nullptr && prior != nullptr && value != nullptr && score != nullptr && header != nullptr && minimax_value != nullptr,
			AGX_ERR_INVALID, "agx_sample_v201_unpack: null argument");
	AGX_REQUIRE(size >= static_cast<size_t>(HEADER_BYTES), AGX_ERR_INVALID, "agx_sample_v201_unpack: %zu bytes are no sample", size);
	uint16_t h16[6];
	uint32_t count;
	std::memcpy(h16, h_bytes, 12);
	std::memcpy(&count, h_bytes + 12, 4);
	AGX_REQUIRE(size >= HEADER_BYTES + static_cast<size_t>(ENTRY_BYTES) * count, AGX_ERR_INVALID, "agx_sample_v201_unpack: truncated sample (%u entries)", count);
	const float value_scale = ScaleFormat::decode(h16[0]), prior_scale = ScaleFormat::decode(h16[1]), visit_scale = ScaleFormat::decode(h16[2]);
	const int hw = rows * cols;
	for (int i = 0; i < hw; i++)
	{ // SearchDataPack::clear: Score() = unknown 0, Value() = (0, 0)
		visits[i] = 0;
		prior[i] = 0.0f;
		value[2 * i] = value[2 * i + 1] = 0.0f;
		score[i] = static_cast<uint16_t>((2u << 13) | 4000u);
	}
	auto valid_value = [](float w, float d, float &ow, float &od)
	{ // get_valid_value (SearchDataStorage.cpp:52-61)
		const float t = w + d;
		if (t > 1.0f)
		{
			w /= t;
			d /= t;
		}
		ow = w;
		od = d;
	};
	int cell = 0, sum_visits = 0;
	float win_rate = 0.0f, draw_rate = 0.0f;
	for (uint32_t k = 0; k < count; k++)
	{ // storeTo (:375-409)
		const uint8_t *q = h_bytes + HEADER_BYTES + ENTRY_BYTES * k;
		cell += q[0];
		AGX_REQUIRE(cell < hw, AGX_ERR_INVALID, "agx_sample_v201_unpack: entry %u lies outside a %dx%d board", k, rows, cols);
		const float v = VisitFormat::decode(q[1]) * visit_scale + 0.5f;
		visits[cell] = static_cast<int>(v);
		float w, d;
		valid_value(PriorFormat::decode(q[4]) * value_scale, PriorFormat::decode(q[5]) * value_scale, w, d);
		value[2 * cell] = w;
		value[2 * cell + 1] = d;
		score[cell] = static_cast<uint16_t>(score_from_code(q[3]));
		prior[cell] = PriorFormat::decode(q[2]) * prior_scale;
		sum_visits = static_cast<int>(static_cast<float>(sum_visits) + v);
		win_rate += w * v;
		draw_rate += d * v;
	}
	header[0] = h16[3];
	header[1] = h16[4];
	header[2] = h16[5];
	if (sum_visits == 0)
	{ // Score::convertToValue of the minimax score (Score.hpp:266-283)
		const uint32_t s = h16[3], pv = (s >> 13) & 3u;
		const int eval = static_cast<int>(s & 8191u) - 4000;
		minimax_value[0] = (pv == 2u) ? (1000 + eval) / 2000.0f : ((pv == 3u && s != 0xFFFFu) ? 1.0f : 0.0f);
		minimax_value[1] = (pv == 1u) ? 1.0f : 0.0f;
	}
	else
		valid_value(win_rate / sum_visits, draw_rate / sum_visits, minimax_value[0], minimax_value[1]);
	if (consumed != nullptr)
		*consumed = HEADER_BYTES + static_cast<size_t>(ENTRY_BYTES) * count;
	return AGX_OK;
}

} /* extern "C" */
