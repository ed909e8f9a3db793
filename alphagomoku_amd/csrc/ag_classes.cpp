/*
 * ag_classes.cpp — the reference-named C++ boundary (the headers under include/alphagomoku_agx) over the C ABI (include/agx.h).
 * Plain host C++ (no HIP types): built by alphagomoku_amd/build.py into libagx_ag.so, linked against libagx.so.
 * Reference lines each class follows are cited in the headers; error behaviour: a non-zero agx status becomes std::logic_error
 * (invalid argument / state) or std::runtime_error (device failure), the exceptions the reference throws (NNEvaluator.cpp:149,185-187).
 */
#include "../../include/alphagomoku_agx/selfplay.hpp"
#include "symmetry.hpp"

#include <algorithm>
#include <chrono>
#include <csignal>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace
{
	void check(int status)
	{
		if (status == AGX_OK)
			return;
		const std::string msg = agx_last_error();
		if (status == AGX_ERR_INVALID || status == AGX_ERR_STATE)
			throw std::logic_error(msg);
		throw std::runtime_error(msg);
	}
	int selector_id(const std::string &policy)
	{ // EdgeSelector::create (EdgeSelector.cpp:680-711): the selectors GameGenerator::make_move can be configured with
		if (policy == "best") return 0;
		if (policy == "max_visit") return 1;
		if (policy == "min_visit") return 2;
		if (policy == "max_value") return 3;
		if (policy == "max_policy") return 4;
		if (policy == "lcb") return 5;
		throw std::logic_error("Unknown selection policy '" + policy + "'");
	}
	int init_to_id(const std::string &s)
	{ // EdgeSelector.cpp:1140-1165
		if (s == "q_head") return 0;
		if (s == "parent") return 1;
		if (s == "draw") return 2;
		if (s == "loss") return 3;
		throw std::logic_error("Unknown init_to '" + s + "'");
	}
	int noise_id(const std::string &s)
	{
		if (s == "none") return 0;
		if (s == "custom") return 1;
		if (s == "dirichlet") return 2;
		if (s == "gumbel") return 3;
		throw std::logic_error("Unknown noise_type '" + s + "'");
	}
	uint64_t mix64(uint64_t z)
	{
		z += 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
}

namespace ag
{
	std::string toString(GameRules rules)
	{
		static const char *names[] = { "FREESTYLE", "STANDARD", "RENJU", "CARO5", "CARO6" };
		return names[static_cast<int>(rules)];
	}
	std::string toString(GameOutcome outcome)
	{
		static const char *names[] = { "UNKNOWN", "DRAW", "CROSS_WIN", "CIRCLE_WIN" };
		return names[static_cast<int>(outcome)];
	}
	std::string TimedStat::toString() const
	{
		const double per = (m_total_count > 0) ? getTotalTime() / m_total_count : 0.0;
		return m_name + " : " + std::to_string(getTotalTime()) + "s : " + std::to_string(m_total_count) + " : " + std::to_string(per * 1.0e6) + " us";
	}

	/* ------------------------------------------------ AGNetwork ------------------------------------------------ */
	AGNetwork::AGNetwork(const GameConfig &gameOptions, const std::string &architecture, int blocks, int filters) :
			game_config(gameOptions)
	{
		if (architecture != "ResnetPV" && architecture != "ResnetPVQ" && architecture != "ResnetPVraw")
			throw std::logic_error("AGNetwork: unknown architecture '" + architecture + "' (the device tower implements ResnetPV, ResnetPVraw and ResnetPVQ)");
		desc.rows = gameOptions.rows;
		desc.cols = gameOptions.cols;
		desc.blocks = blocks;
		desc.filters = filters;
		desc.in_channels = (architecture == "ResnetPVraw") ? 8 : 32; // networks.cpp:107-129: the raw network sees the 8 low bits of a feature word
		desc.value_hidden = std::min(256, 2 * filters);
		desc.action_values = (architecture == "ResnetPVQ") ? 1 : 0;
		create();
	}
	AGNetwork::~AGNetwork()
	{
		release();
	}
	void AGNetwork::create()
	{
		check(agx_net_create(&desc, &net));
		if (!blob_copy.empty())
			check(agx_net_load_weights(net, blob_copy.data(), blob_copy.size()));
	}
	void AGNetwork::release()
	{
		void **buffers[] = { &d_input, &d_policy, &d_value, &d_action_values };
		for (void **p : buffers)
		{
			if (*p != nullptr)
				agx_free(*p);
			*p = nullptr;
		}
		if (net != nullptr)
			agx_net_destroy(net);
		net = nullptr;
	}
	std::string AGNetwork::getOutputConfig() const
	{
		return desc.action_values ? "pvq" : "pv";
	}
	std::string AGNetwork::name() const
	{
		return desc.action_values ? "ResnetPVQ" : (desc.in_channels == 8 ? "ResnetPVraw" : "ResnetPV");
	}
	size_t AGNetwork::numberOfWeights() const
	{
		return agx_net_blob_floats(&desc);
	}
	void AGNetwork::loadWeights(const std::vector<float> &blob)
	{
		if (net == nullptr)
			throw std::logic_error("AGNetwork::loadWeights() : the network has not been created");
		check(agx_net_load_weights(net, blob.data(), blob.size()));
		blob_copy = blob;
	}
	void AGNetwork::packInputData(int index, const uint32_t *features)
	{
		if (index < 0 || index >= batch_size)
			throw std::logic_error("AGNetwork::packInputData() : index " + std::to_string(index) + " outside the batch of " + std::to_string(batch_size));
		const int hw = desc.rows * desc.cols;
		std::memcpy(input.data() + static_cast<size_t>(index) * hw, features, sizeof(uint32_t) * hw);
	}
	void AGNetwork::unpackOutput(int index, std::vector<float> &out_policy, std::vector<Value> &actionValues, Value &out_value, float &movesLeft) const
	{ // NetworkDataPack::unpackPolicy / unpackValue / unpackActionValues (NetworkDataPack.cpp:199-236)
		if (index < 0 || index >= batch_size)
			throw std::logic_error("AGNetwork::unpackOutput() : index outside the batch");
		const int hw = desc.rows * desc.cols;
		out_policy.assign(policy.begin() + static_cast<size_t>(index) * hw, policy.begin() + static_cast<size_t>(index + 1) * hw);
		actionValues.assign(hw, Value());
		if (desc.action_values)
			for (int i = 0; i < hw; i++)
				actionValues[i] = Value(action_values[(static_cast<size_t>(index) * hw + i) * 2], action_values[(static_cast<size_t>(index) * hw + i) * 2 + 1]);
		out_value = Value(value[3 * index], value[3 * index + 1]); // (win, draw) of the softmax over (win, draw, loss)
		movesLeft = 0.0f;                                            // 'pv' / 'pvq' networks have no moves-left output
	}
	void AGNetwork::asyncForwardLaunch(int batch)
	{
		if (!isLoaded())
			throw std::logic_error("AGNetwork::forward() : the network has not been loaded");
		if (batch < 0 || batch > batch_size)
			throw std::logic_error("AGNetwork::forward() : batch " + std::to_string(batch) + " exceeds the batch size " + std::to_string(batch_size));
		const int hw = desc.rows * desc.cols;
		check(agx_memcpy_h2d(d_input, input.data(), sizeof(uint32_t) * hw * batch));
		if (desc.action_values)
			check(agx_nn_forward_pvq(net, static_cast<const uint32_t*>(d_input), batch, static_cast<float*>(d_policy), static_cast<float*>(d_value),
					static_cast<float*>(d_action_values), nullptr));
		else
			check(agx_nn_forward(net, static_cast<const uint32_t*>(d_input), batch, static_cast<float*>(d_policy), static_cast<float*>(d_value), nullptr));
		launched = batch;
	}
	void AGNetwork::asyncForwardJoin()
	{
		const int hw = desc.rows * desc.cols;
		check(agx_device_synchronize());
		check(agx_memcpy_d2h(policy.data(), d_policy, sizeof(float) * hw * launched));
		check(agx_memcpy_d2h(value.data(), d_value, sizeof(float) * 3 * launched));
		if (desc.action_values)
			check(agx_memcpy_d2h(action_values.data(), d_action_values, sizeof(float) * 2 * hw * launched));
		launched = 0;
	}
	void AGNetwork::forward(int batch)
	{
		asyncForwardLaunch(batch);
		asyncForwardJoin();
	}
	void AGNetwork::optimize(int)
	{
	}
	void AGNetwork::convertToHalfFloats()
	{
	}
	void AGNetwork::saveToFile(const std::string &path) const
	{
		if (blob_copy.empty())
			throw std::logic_error("AGNetwork::saveToFile() : no weights loaded");
		std::ofstream f(path, std::ofstream::binary);
		if (!f.good())
			throw std::runtime_error("AGNetwork::saveToFile() : cannot open '" + path + "'");
		const uint64_t count = blob_copy.size();
		f.write("AGXW", 4);
		f.write(reinterpret_cast<const char*>(&desc), sizeof(desc));
		f.write(reinterpret_cast<const char*>(&count), sizeof(count));
		f.write(reinterpret_cast<const char*>(blob_copy.data()), sizeof(float) * count);
	}
	void AGNetwork::loadFromFile(const std::string &path)
	{
		std::ifstream f(path, std::ifstream::binary);
		if (!f.good())
			throw std::runtime_error("File '" + path + "' does not exist"); // FileLoader (file_util.cpp:58-60)
		char magic[4];
		uint64_t count = 0;
		AgxNetDesc d;
		f.read(magic, 4);
		f.read(reinterpret_cast<char*>(&d), sizeof(d));
		f.read(reinterpret_cast<char*>(&count), sizeof(count));
		if (!f.good() || std::memcmp(magic, "AGXW", 4) != 0)
			throw std::runtime_error("'" + path + "' is not a weight file of this library (MinML checkpoints cannot be imported: their format is not in the reference tree)");
		std::vector<float> blob(count);
		f.read(reinterpret_cast<char*>(blob.data()), sizeof(float) * count);
		if (!f.good())
			throw std::runtime_error("'" + path + "' is truncated");
		release();
		desc = d;
		game_config.rows = d.rows;
		game_config.cols = d.cols;
		if (game_config.draw_after <= 0)
			game_config.draw_after = d.rows * d.cols;
		blob_copy.clear();
		create();
		loadWeights(blob);
		if (batch_size > 0)
		{
			const int b = batch_size;
			batch_size = 0;
			setBatchSize(b);
		}
	}
	void AGNetwork::unloadGraph()
	{
		release();
	}
	bool AGNetwork::isLoaded() const noexcept
	{
		return net != nullptr && !blob_copy.empty();
	}
	void AGNetwork::synchronize()
	{
		check(agx_device_synchronize());
	}
	void AGNetwork::moveTo(Device device)
	{ // the device copy lives where it was created: re-create it on the target device
		if (device.isCPU())
			throw std::logic_error("AGNetwork::moveTo() : there is no CPU path");
		const int b = batch_size;
		release();
		batch_size = 0;
		check(agx_set_device(device.index()));
		create();
		if (b > 0)
			setBatchSize(b);
	}
	int AGNetwork::getBatchSize() const noexcept
	{
		return batch_size;
	}
	void AGNetwork::setBatchSize(int batchSize)
	{
		if (batchSize <= 0)
			throw std::logic_error("AGNetwork::setBatchSize() : batch size must be positive");
		if (batchSize == batch_size && d_input != nullptr)
			return;
		void **buffers[] = { &d_input, &d_policy, &d_value, &d_action_values };
		for (void **p : buffers)
		{
			if (*p != nullptr)
				agx_free(*p);
			*p = nullptr;
		}
		batch_size = batchSize;
		const size_t hw = static_cast<size_t>(desc.rows) * desc.cols, n = batchSize;
		input.assign(n * hw, 0u);
		policy.assign(n * hw, 0.0f);
		value.assign(n * 3, 0.0f);
		action_values.assign(desc.action_values ? n * hw * 2 : 0, 0.0f);
		check(agx_malloc(&d_input, sizeof(uint32_t) * n * hw));
		check(agx_malloc(&d_policy, sizeof(float) * n * hw));
		check(agx_malloc(&d_value, sizeof(float) * n * 3));
		if (desc.action_values)
			check(agx_malloc(&d_action_values, sizeof(float) * n * hw * 2));
	}
	GameConfig AGNetwork::getGameConfig() const noexcept
	{
		return game_config;
	}
	std::unique_ptr<AGNetwork> loadAGNetwork(const std::string &path)
	{
		std::unique_ptr<AGNetwork> result = std::make_unique<AGNetwork>();
		result->loadFromFile(path);
		return result;
	}

	NetworkLoader::NetworkLoader(const char *path) :
			NetworkLoader(std::string(path))
	{
	}
	NetworkLoader::NetworkLoader(const std::string &path) :
			paths( { path })
	{
	}
	NetworkLoader::NetworkLoader(const std::vector<std::string> &path) :
			paths(path)
	{
	}
	std::unique_ptr<AGNetwork> NetworkLoader::get(bool) const
	{ // NetworkLoader.cpp:44-56
		if (paths.empty())
			return nullptr;
		std::unique_ptr<AGNetwork> result = loadAGNetwork(paths.at(0));
		if (paths.size() > 1)
		{ // running average of the weights (ml::averageModelWeights(alpha, network, 1 - alpha, result))
			std::vector<float> sum;
			{
				std::ifstream f(paths[0], std::ifstream::binary);
				f.seekg(4 + sizeof(AgxNetDesc) + sizeof(uint64_t));
				sum.resize(result->numberOfWeights());
				f.read(reinterpret_cast<char*>(sum.data()), sizeof(float) * sum.size());
			}
			for (size_t i = 1; i < paths.size(); i++)
			{
				std::ifstream f(paths[i], std::ifstream::binary);
				if (!f.good())
					throw std::runtime_error("File '" + paths[i] + "' does not exist");
				f.seekg(4 + sizeof(AgxNetDesc) + sizeof(uint64_t));
				std::vector<float> w(sum.size());
				f.read(reinterpret_cast<char*>(w.data()), sizeof(float) * w.size());
				const float alpha = 1.0f / (i + 1);
				for (size_t k = 0; k < sum.size(); k++)
					sum[k] = alpha * w[k] + (1.0f - alpha) * sum[k];
			}
			result->loadWeights(sum);
		}
		return result;
	}

	/* ------------------------------------------------ NNEvaluator ------------------------------------------------ */
	NNEvaluatorStats::NNEvaluatorStats() :
			pack("pack   "), compute("compute"), unpack("unpack ")
	{
	}
	std::string NNEvaluatorStats::toString() const
	{
		std::string result = "----NNEvaluator----\n";
		result += "total samples = " + std::to_string(batch_sizes) + '\n';
		result += pack.toString() + '\n' + compute.toString() + '\n' + unpack.toString() + '\n';
		return result;
	}
	NNEvaluatorStats& NNEvaluatorStats::operator+=(const NNEvaluatorStats &other) noexcept
	{
		batch_sizes += other.batch_sizes;
		pack += other.pack;
		compute += other.compute;
		unpack += other.unpack;
		return *this;
	}
	NNEvaluatorStats& NNEvaluatorStats::operator/=(int i) noexcept
	{
		batch_sizes /= std::max(1, i);
		return *this;
	}

	NNEvaluator::NNEvaluator(const DeviceConfig &cfg) :
			config(cfg)
	{
		if (cfg.device.isCPU())
			throw std::logic_error("NNEvaluator : the device engine has no CPU path, DeviceConfig::device must name a GPU");
	}
	bool NNEvaluator::isOnGPU() const noexcept
	{
		return true;
	}
	void NNEvaluator::clearStats() noexcept
	{
		stats = NNEvaluatorStats();
	}
	NNEvaluatorStats NNEvaluator::getStats() const noexcept
	{
		return stats;
	}
	bool NNEvaluator::isQueueFull() const noexcept
	{ // NNEvaluator.cpp:99-102; a scheduled pool slice is a full launch by itself
		return !waiting_slices.empty() || (network != nullptr && static_cast<int>(waiting_queue.size()) >= network->getBatchSize());
	}
	int NNEvaluator::getQueueSize() const noexcept
	{
		int result = static_cast<int>(waiting_queue.size());
		for (const SliceData &s : waiting_slices)
			result += s.positions;
		return result;
	}
	void NNEvaluator::clearQueue() noexcept
	{
		waiting_queue.clear();
		waiting_slices.clear();
	}
	void NNEvaluator::useSymmetries(bool b) noexcept
	{
		use_symmetries = b;
	}
	void NNEvaluator::loadGraph(const NetworkLoader &loader)
	{ // NNEvaluator.cpp:121-129
		network = loader.get();
		if (network == nullptr)
			throw std::logic_error("NNEvaluator::loadGraph() : the loader holds no network");
		get_network().optimize(2);
		get_network().moveTo(config.device);
		// host-side staging for externally owned tasks only (pool slices never leave the device): DeviceConfig::batch_size beyond 1024
		// positions is a slicing rule for the pool, not a staging size
		get_network().setBatchSize(std::max(1, std::min(config.batch_size, 1024)));
		get_network().convertToHalfFloats();
		get_network().forward(1);
	}
	void NNEvaluator::unloadGraph()
	{
		get_network().unloadGraph();
	}
	void NNEvaluator::addToQueue(SearchTask &task)
	{ // NNEvaluator.cpp:134-141; the reference draws randInt(8) from a time-seeded generator, here a counter-based hash
		if (use_symmetries)
			waiting_queue.push_back( { &task, static_cast<int>(mix64(0x5DEECE66Dull ^ symmetry_counter++) >> 61) });
		else
			waiting_queue.push_back( { &task, 0 });
	}
	void NNEvaluator::addToQueue(SearchTask &task, int symmetry)
	{
		if (symmetry < 0 || symmetry >= 8)
			throw std::logic_error("NNEvaluator::addToQueue() : symmetry must be in [0, 8)");
		waiting_queue.push_back( { &task, symmetry });
	}
	void NNEvaluator::addToQueue(AgxEngine *engine, int group, int n_groups, int max_positions, void *stream, bool *ready)
	{
		SliceData s;
		s.engine = engine;
		s.group = group;
		s.n_groups = n_groups;
		s.positions = max_positions;
		s.stream = stream;
		s.ready_flag = ready;
		waiting_slices.push_back(s);
	}
	void NNEvaluator::addToQueueOverlapped(AgxEngine *engine, int buffer, int max_positions, void *stream, bool *ready)
	{
		if (own_stream == nullptr)
		{
			check(agx_stream_create(&own_stream));
			for (int i = 0; i < 2; i++)
			{
				check(agx_event_create(&scheduled_event[i]));
				check(agx_event_create(&done_event[i]));
			}
		}
		SliceData s;
		s.engine = engine;
		s.group = buffer;
		s.n_groups = 2;
		s.positions = max_positions;
		s.stream = stream;
		s.ready_flag = ready;
		s.overlap = true;
		s.event = buffer;
		waiting_slices.push_back(s);
	}
	NNEvaluator::~NNEvaluator()
	{
		if (own_stream != nullptr)
		{
			agx_stream_synchronize(own_stream);
			for (int i = 0; i < 2; i++)
			{
				agx_event_destroy(scheduled_event[i]);
				agx_event_destroy(done_event[i]);
				if (network_timer[i] != nullptr)
					agx_timer_destroy(network_timer[i]);
			}
			agx_stream_destroy(own_stream);
		}
	}
	double NNEvaluator::evaluateGraph()
	{ // NNEvaluator.cpp:147-181
		if (!get_network().isLoaded())
			throw std::logic_error("graph is empty - the network has not been loaded");
		const auto t0 = std::chrono::steady_clock::now();
		uint64_t samples = 0;
		while (!waiting_queue.empty() || !waiting_slices.empty())
		{
			samples += std::min<size_t>(waiting_queue.size(), get_network().getBatchSize());
			asyncEvaluateGraphLaunch();
			asyncEvaluateGraphJoin();
		}
		const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		return (samples > 0) ? seconds / samples : 0.0;
	}
	double NNEvaluator::asyncEvaluateGraphLaunch()
	{ // NNEvaluator.cpp:182-206
		if (!get_network().isLoaded())
			throw std::logic_error("graph is empty - the network has not been loaded");
		if (!in_progress_queue.empty() || !in_progress_slices.empty())
			throw std::logic_error("some tasks are already being processed");
		// pool slices: the batch is the slice's device-side queue, the launch goes onto the slice's stream behind its solver kernel
		in_progress_slices.swap(waiting_slices);
		bool timed_launch = false;
		for (const SliceData &s : in_progress_slices)
			if (s.overlap)
			{ // behind everything the search stream holds so far (select + solve of this buffer), beside whatever it is given next
				check(agx_event_record(scheduled_event[s.event], s.stream));
				check(agx_stream_wait_event(own_stream, scheduled_event[s.event]));
				// the estimate asyncEvaluateGraphLaunch returns (PerfEstimator in the reference, NNEvaluator.cpp:197-206): device time of this
				// buffer's PREVIOUS network launch (long finished: the search stream has waited for it), smoothed
				if (network_timer[s.event] == nullptr)
					check(agx_timer_create(&network_timer[s.event]));
				else if (network_timer_used[s.event])
				{
					float ms = 0.0f;
					int ready = 0; // (polled, never waited for: a launch that is still running keeps the previous estimate)
					if (agx_timer_poll_ms(network_timer[s.event], &ms, &ready) == AGX_OK && ready != 0 && ms > 0.0f)
						network_seconds = (network_seconds > 0.0) ? 0.75 * network_seconds + 0.25 * ms * 1.0e-3 : ms * 1.0e-3;
				}
				check(agx_timer_start(network_timer[s.event], own_stream));
				check(agx_engine_evaluate_group(s.engine, get_network().handle(), s.group, s.n_groups, own_stream));
				check(agx_timer_stop(network_timer[s.event], own_stream));
				network_timer_used[s.event] = true;
				timed_launch = true;
				check(agx_event_record(done_event[s.event], own_stream));
			}
			else
				check(agx_engine_evaluate_group(s.engine, get_network().handle(), s.group, s.n_groups, s.stream));
		// host-side tasks
		const int batch = std::min(static_cast<int>(waiting_queue.size()), get_network().getBatchSize());
		if (batch > 0)
		{
			in_progress_queue.assign(waiting_queue.begin(), waiting_queue.begin() + batch);
			waiting_queue.erase(waiting_queue.begin(), waiting_queue.begin() + batch);
			pack_to_network();
			get_network().asyncForwardLaunch(batch);
		}
		stats.compute.startTimer();
		// the estimated end of the launch (NNEvaluator.cpp:206; SearchThread::asynchronous_run hands it to Search::solve as its deadline); a negative
		// value — no estimate yet, or a pool slice whose launches nobody times — makes Search::solve run its ordinary node budget
		if (timed_launch && network_seconds > 0.0)
			return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() + network_seconds;
		return -1.0;
	}
	void NNEvaluator::asyncEvaluateGraphJoin()
	{ // NNEvaluator.cpp:207-228
		if (!get_network().isLoaded())
			throw std::logic_error("graph is empty - the network has not been loaded");
		if (in_progress_queue.empty() && in_progress_slices.empty())
			return;
		const int batch = static_cast<int>(in_progress_queue.size());
		if (batch > 0)
		{
			get_network().asyncForwardJoin();
			stats.batch_sizes += batch;
			unpack_from_network();
			in_progress_queue.clear();
		}
		// a slice's expand kernel is enqueued on the same stream as its network launch, so "joined" needs no host wait; an overlapped
		// launch is joined on the device: the search stream waits for the network's event before what the caller enqueues next
		for (const SliceData &s : in_progress_slices)
		{
			if (s.overlap)
				check(agx_stream_wait_event(s.stream, done_event[s.event]));
			if (s.ready_flag != nullptr)
				*s.ready_flag = true;
		}
		stats.compute.stopTimer(static_cast<int>(in_progress_slices.size()) + (batch > 0 ? 1 : 0));
		in_progress_slices.clear();
	}
	AGNetwork& NNEvaluator::get_network()
	{
		if (network == nullptr)
			throw std::logic_error("NNEvaluator::get_network() : network has not been initialized");
		return *network;
	}
	const AGNetwork& NNEvaluator::get_network() const
	{
		if (network == nullptr)
			throw std::logic_error("NNEvaluator::get_network() : network has not been initialized");
		return *network;
	}
	void NNEvaluator::pack_to_network()
	{ // NNEvaluator.cpp:244-262: features.augment(symmetry) + packInputData
		stats.pack.startTimer();
		const GameConfig cfg = get_network().getGameConfig();
		const int n = cfg.rows, hw = cfg.rows * cfg.cols;
		std::vector<uint32_t> tmp(hw);
		for (size_t i = 0; i < in_progress_queue.size(); i++)
		{
			const TaskData td = in_progress_queue[i];
			const std::vector<uint32_t> &f = td.ptr->getFeatures();
			if (static_cast<int>(f.size()) != hw)
				throw std::logic_error("NNEvaluator : task of another board size");
			for (int r = 0; r < n; r++)
				for (int c = 0; c < n; c++)
				{
					int sr, sc;
					agx::symmetry_source(td.symmetry, n, r, c, sr, sc);
					tmp[r * n + c] = agx::shuffle_feature_directions(f[sr * n + sc], td.symmetry);
				}
			get_network().packInputData(static_cast<int>(i), tmp.data());
		}
		stats.pack.stopTimer(static_cast<int>(in_progress_queue.size()));
	}
	void NNEvaluator::unpack_from_network()
	{ // NNEvaluator.cpp:263-286
		stats.unpack.startTimer();
		const GameConfig cfg = get_network().getGameConfig();
		const int n = cfg.rows;
		std::vector<float> policy;
		std::vector<Value> action_values;
		Value value;
		float moves_left = 0.0f;
		for (size_t i = 0; i < in_progress_queue.size(); i++)
		{
			const TaskData td = in_progress_queue[i];
			get_network().unpackOutput(static_cast<int>(i), policy, action_values, value, moves_left);
			const int inv = agx::inverse_symmetry(td.symmetry);
			for (int r = 0; r < n; r++)
				for (int c = 0; c < n; c++)
				{
					int sr, sc;
					agx::symmetry_source(inv, n, r, c, sr, sc);
					td.ptr->getPolicy()[r * n + c] = policy[sr * n + sc];
					td.ptr->getActionValues()[r * n + c] = action_values[sr * n + sc];
				}
			td.ptr->setValue(value);
			if (td.ptr->getScore().isUnproven())
				td.ptr->setMovesLeft(moves_left);
			td.ptr->markAsProcessedByNetwork();
		}
		stats.unpack.stopTimer(static_cast<int>(in_progress_queue.size()));
	}

	/* ------------------------------------------------ Search / Tree ------------------------------------------------ */
	SearchStats::SearchStats() :
			select("select  "), solve("solve   "), schedule("schedule"), generate("generate"), expand("expand  "), backup("backup  ")
	{
	}
	std::string SearchStats::toString() const
	{
		std::string result = "----SearchStats----\n";
		result += "nb_duplicate_nodes     = " + std::to_string(nb_duplicate_nodes) + '\n';
		result += "nb_information_leaks   = " + std::to_string(nb_information_leaks) + '\n';
		result += "nb_wasted_expansions   = " + std::to_string(nb_wasted_expansions) + '\n';
		result += "nb_proven_states       = " + std::to_string(nb_proven_states) + '\n';
		result += "nb_network_evaluations = " + std::to_string(nb_network_evaluations) + '\n';
		result += "nb_node_count          = " + std::to_string(nb_node_count) + '\n';
		result += select.toString() + '\n' + solve.toString() + '\n' + expand.toString() + '\n' + backup.toString() + '\n';
		return result;
	}
	SearchStats& SearchStats::operator+=(const SearchStats &other) noexcept
	{
		select += other.select;
		solve += other.solve;
		schedule += other.schedule;
		generate += other.generate;
		expand += other.expand;
		backup += other.backup;
		nb_duplicate_nodes += other.nb_duplicate_nodes;
		nb_information_leaks += other.nb_information_leaks;
		nb_wasted_expansions += other.nb_wasted_expansions;
		nb_proven_states += other.nb_proven_states;
		nb_network_evaluations += other.nb_network_evaluations;
		nb_node_count += other.nb_node_count;
		return *this;
	}
	SearchStats& SearchStats::operator/=(int i) noexcept
	{
		i = std::max(1, i);
		nb_duplicate_nodes /= i;
		nb_information_leaks /= i;
		nb_wasted_expansions /= i;
		nb_proven_states /= i;
		nb_network_evaluations /= i;
		nb_node_count /= i;
		return *this;
	}
	double SearchStats::getTotalTime() const noexcept
	{
		return select.getTotalTime() + solve.getTotalTime() + schedule.getTotalTime() + generate.getTotalTime() + expand.getTotalTime() + backup.getTotalTime();
	}
	std::string NodeCacheStats::toString() const
	{
		return "----NodeCacheStats----\nstored nodes (peak per game) = " + std::to_string(stored_nodes) + "\nstored edges (peak per game) = " + std::to_string(stored_edges) + '\n';
	}
	NodeCacheStats& NodeCacheStats::operator+=(const NodeCacheStats &other) noexcept
	{
		stored_nodes = std::max(stored_nodes, other.stored_nodes);
		stored_edges = std::max(stored_edges, other.stored_edges);
		return *this;
	}
	NodeCacheStats& NodeCacheStats::operator/=(int) noexcept
	{
		return *this;
	}

	GamePool::GamePool(const GameConfig &gameOptions, const SearchConfig &searchOptions, const EdgeSelectorConfig &finalSelector, int games, int maxSimulations,
			bool useSymmetries, const std::string &networkOutputs, bool forceExpandRoot, int searchBuffers) :
			game_config(gameOptions), search_config(searchOptions), games(games), batch(searchOptions.max_batch_size)
	{
		if (searchBuffers == 2 && games != 1)
			throw std::logic_error("GamePool : two task buffers are for a pool of ONE tree");
		if (gameOptions.rows != gameOptions.cols)
			throw std::logic_error("GamePool : only square boards are supported");
		AgxEngineConfig c;
		check(agx_engine_default_config(&c));
		c.rules = static_cast<int>(gameOptions.rules);
		c.board_size = gameOptions.rows;
		c.draw_after = gameOptions.draw_after;
		c.n_games = games;
		c.max_batch_size = searchOptions.max_batch_size;
		c.max_simulations = maxSimulations;
		const EdgeSelectorConfig &sel = searchOptions.mcts_config.edge_selector_config;
		if (sel.policy != "puct")
			throw std::logic_error("GamePool : the device search implements the 'puct' selector (EdgeSelectorConfig::policy = '" + sel.policy + "')");
		c.exploration_constant = sel.exploration_constant;
		c.exploration_scaling = sel.exploration_scaling;
		c.init_to = init_to_id(sel.init_to);
		c.noise_type = noise_id(sel.noise_type);
		c.noise_weight = sel.noise_weight;
		c.information_leak_threshold = searchOptions.tree_config.information_leak_threshold;
		c.node_capacity = searchOptions.tree_config.node_bucket_size;
		c.edge_capacity = searchOptions.tree_config.edge_bucket_size;
		c.policy_expansion_threshold = searchOptions.mcts_config.policy_expansion_threshold;
		c.policy_temperature = searchOptions.mcts_config.policy_temperature;
		c.max_children = (searchOptions.mcts_config.max_children == std::numeric_limits<int>::max()) ? 0 : searchOptions.mcts_config.max_children;
		c.tss_max_positions = searchOptions.tss_config.max_positions;
		c.tss_table_entries = static_cast<uint64_t>(searchOptions.tss_config.hash_table_size);
		c.final_selector = selector_id(finalSelector.policy);
		c.use_symmetries = useSymmetries ? 1 : 0;
		c.action_values = (networkOutputs == "pvq") ? 1 : 0;
		c.record_format = 2;            // samples leave the device in dataset format 201
		c.force_expand_root = forceExpandRoot ? 1 : 0;
		// pacing only, per-game results do not depend on either: the leaves of a batch solved in parallel (k_search_spec), stragglers of a
		// launch put off to the next one
		c.speculative_solver = 1;
		c.solver_yield_fraction = (games >= 64) ? 0.5f : 0.0f;   // (swept 0.3 .. 0.8 on the 1024-game pool, round 5 with 16 solver waves per unit: 0.4-0.5 best)
		if (searchBuffers == 2)
		{ // records 0 and 1 are the two task buffers of the one tree (stepped as group 0 / 1 of 2)
			c.n_games = 2;
			c.search_buffers = 2;
			this->games = 2;
		}
		check(agx_engine_create(&c, &engine));
	}
	GamePool::~GamePool()
	{
		agx_engine_destroy(engine);
	}
	void GamePool::begin(const std::vector<uint16_t> &openings)
	{
		if (openings.empty() || openings.size() % AGX_OPENING_CAP != 0)
			throw std::logic_error("GamePool::begin() : openings must hold a positive multiple of AGX_OPENING_CAP words");
		check(agx_engine_begin(engine, openings.data(), static_cast<int>(openings.size() / AGX_OPENING_CAP), nullptr));
		check(agx_device_synchronize());
	}
	void GamePool::addOpenings(const std::vector<uint16_t> &openings)
	{
		check(agx_engine_add_openings(engine, openings.data(), static_cast<int>(openings.size() / AGX_OPENING_CAP)));
	}
	AgxEngineStats GamePool::getStats() const
	{
		AgxEngineStats s;
		check(agx_device_synchronize());
		check(agx_engine_stats(engine, &s));
		return s;
	}

	Tree::Tree(const TreeConfig &treeConfig) :
			config(treeConfig), standalone(true)
	{ // its storage is the one-game engine of the Search it will be used with (Search::bind)
	}
	Tree::Tree(GamePool &pool, int group, int n_groups, void *stream) :
			pool(&pool), group(group), n_groups(n_groups), stream(stream)
	{
		const int per = (pool.numberOfGames() + n_groups - 1) / n_groups;
		first_game = group * per;
		game_count = std::min(per, pool.numberOfGames() - first_game);
		if (game_count <= 0)
			throw std::logic_error("Tree : slice " + std::to_string(group) + " of " + std::to_string(n_groups) + " is empty");
	}
	GamePool& Tree::bound() const
	{
		if (pool == nullptr)
			throw std::logic_error("Tree : a tree made from a TreeConfig holds its nodes in the engine of a Search: pass it to that Search first (Search::cleanup / select)");
		return *pool;
	}
	void Tree::setBoard(const matrix<Sign> &newBoard, Sign signToMove, bool forceRemoveRootNode)
	{ // Tree.cpp:128-151 on the device: the cached states reachable from the new position stay, the root is the cached node of the position
		GamePool &p = bound();
		if (!standalone)
			throw std::logic_error("Tree::setBoard() : the trees of a pool slice take their positions from the pool's own games");
		if (forceRemoveRootNode)
			throw std::logic_error("Tree::setBoard() : forceRemoveRootNode is not provided by the device tree");
		if (newBoard.rows() != p.getGameConfig().rows || newBoard.cols() != p.getGameConfig().cols)
			throw std::logic_error("Tree::setBoard() : the board is " + std::to_string(newBoard.rows()) + "x" + std::to_string(newBoard.cols()) + ", the search was created for "
					+ std::to_string(p.getGameConfig().rows) + "x" + std::to_string(p.getGameConfig().cols));
		if (signToMove != Sign::CROSS && signToMove != Sign::CIRCLE)
			throw std::logic_error("Tree::setBoard() : signToMove must be CROSS or CIRCLE");
		std::vector<uint8_t> cells(static_cast<size_t>(newBoard.size()));
		for (int i = 0; i < newBoard.size(); i++)
			cells[i] = static_cast<uint8_t>(newBoard[i]);
		check(agx_engine_set_board(p.handle(), 0, cells.data(), static_cast<int>(signToMove), stream));
		summary_valid = false;
		edge_selector.reset();  // a fresh selector / generator per position, as Player::setBoard installs them
		edge_generator.reset();
	}
	void Tree::setEdgeSelector(const EdgeSelector &selector)
	{ // the device's select stage is the PUCT selector with the parameters the engine was created with: accept exactly that
		const EdgeSelectorConfig &want = selector.getConfig(), &have = bound().getSearchConfig().mcts_config.edge_selector_config;
		if (want.policy != "puct")
			throw std::logic_error("Tree::setEdgeSelector() : the device search implements the 'puct' selector, not '" + want.policy + "'");
		if (want.init_to != have.init_to || want.noise_type != have.noise_type || want.noise_weight != have.noise_weight
				|| want.exploration_constant != have.exploration_constant || want.exploration_scaling != have.exploration_scaling)
			throw std::logic_error("Tree::setEdgeSelector() : the selector's parameters differ from the SearchConfig the search was created with");
		edge_selector = selector.clone();
	}
	void Tree::setEdgeGenerator(const EdgeGenerator &generator)
	{
		const UnifiedGenerator *g = dynamic_cast<const UnifiedGenerator*>(&generator);
		if (g == nullptr)
			throw std::logic_error("Tree::setEdgeGenerator() : the device's expand stage implements UnifiedGenerator");
		const MCTSConfig &have = bound().getSearchConfig().mcts_config;
		if (g->maxEdges() != have.max_children || g->expansionThreshold() != have.policy_expansion_threshold || g->policyTemperature() != have.policy_temperature)
			throw std::logic_error("Tree::setEdgeGenerator() : the generator's parameters differ from the SearchConfig the search was created with");
		if (standalone)
			check(agx_engine_set_force_expand_root(bound().handle(), g->forceExpandRoot() ? 1 : 0)); // false for a Player (Player.cpp:109), true in self-play
		else if (!g->forceExpandRoot())
			throw std::logic_error("Tree::setEdgeGenerator() : the trees of a self-play pool never prune the root (forceExpandRoot = true, GameGenerator.cpp:183-184)");
		edge_generator = generator.clone();
	}
	int64_t Tree::getMemory() const noexcept
	{
		return 0;
	}
	namespace
	{
		AgxGameInfo game_info(const GamePool &pool, int game, std::vector<AgxEdgeView> *edges = nullptr, std::vector<uint8_t> *board = nullptr)
		{
			AgxGameInfo info;
			std::vector<AgxEdgeView> e(400);
			std::vector<uint8_t> b(400);
			check(agx_engine_game_info(pool.handle(), game, &info, b.data(), edges ? e.data() : nullptr, 400));
			if (edges != nullptr)
				edges->assign(e.begin(), e.begin() + info.root_edges);
			if (board != nullptr)
				*board = b;
			return info;
		}
	}
	namespace
	{
	}
	const int* Tree::root_summary() const
	{ // one read serves the getters of a stop condition; any stage that changes the tree (Search::expand, cleanup, Tree::setBoard) drops it
		if (!summary_valid)
		{
			check(agx_engine_root_summary(bound().handle(), 0, stream, summary));
			summary_valid = true;
		}
		return summary;
	}
	int Tree::getSimulationCount(int game) const
	{
		if (standalone)
			return root_summary()[0];
		return game_info(bound(), first_game + game).root_visits;
	}
	bool Tree::isRootProven(int game) const
	{
		if (standalone)
			return root_summary()[1] != 0;
		return Score::from_short(static_cast<uint16_t>(game_info(bound(), first_game + game).root_score)).isProven();
	}
	int Tree::getNodeCount(int game) const
	{
		if (standalone)
			return root_summary()[2];
		return game_info(bound(), first_game + game).n_nodes;
	}
	int Tree::getMoveNumber(int game) const
	{
		return game_info(bound(), first_game + game).n_moves;
	}
	Value Tree::getEvaluation(int game) const
	{
		const AgxGameInfo info = game_info(bound(), first_game + game);
		return Value(info.root_win, info.root_draw);
	}
	float Tree::getExpectation(int game) const
	{
		return getEvaluation(game).getExpectation();
	}
	Sign Tree::getSignToMove(int game) const
	{
		return static_cast<Sign>(game_info(bound(), first_game + game).sign_to_move);
	}
	matrix<Sign> Tree::getBoard(int game) const
	{
		std::vector<uint8_t> b;
		game_info(bound(), first_game + game, nullptr, &b);
		const GameConfig &gc = bound().getGameConfig();
		matrix<Sign> result(gc.rows, gc.cols);
		for (int i = 0; i < result.size(); i++)
			result[i] = static_cast<Sign>(b[i]);
		return result;
	}
	const matrix<Sign>& Tree::getBoard() const
	{
		board_copy = getBoard(0);
		return board_copy;
	}
	void Tree::clear()
	{ // Tree.cpp:124-127.  Both callers of the reference empty the solver's table in the same breath (GameGenerator.cpp:52-53, SearchEngine.cpp:91-96)
	  // and give the tree its position next; the games of a pool slice restart on the device by themselves
		if (!standalone)
			throw std::logic_error("Tree::clear() : the trees of a pool slice are cleared on the device when their games restart");
		if (pool == nullptr)
			return; // not bound to a Search yet: nothing stored
		pool->begin(std::vector<uint16_t>(AGX_OPENING_CAP, 0));
		summary_valid = false;
	}
	float Tree::getMovesLeft(int game) const
	{
		return game_info(bound(), first_game + game).root_moves_left;
	}
	int Tree::getMaximumDepth(int game) const
	{
		return game_info(bound(), first_game + game).max_depth;
	}
	bool Tree::hasAllMovesProven(int game) const
	{ // Tree.cpp:199-206 (std::all_of over the root's edges: true for a root without edges, false without a root)
		const Node root = getInfo(game);
		if (root.getVisits() == 0 && root.numberOfEdges() == 0)
			return false;
		for (const Edge *e = root.begin(); e < root.end(); e++)
			if (!e->getScore().isProven())
				return false;
		return true;
	}
	bool Tree::hasSingleMove(int game) const
	{ // Tree.cpp:207-213
		return game_info(bound(), first_game + game).root_edges == 1;
	}
	bool Tree::hasSingleNonLosingMove(int game) const
	{ // Tree.cpp:214-224
		const Node root = getInfo(game);
		int non_losing = 0;
		for (const Edge *e = root.begin(); e < root.end(); e++)
			non_losing += (e->getScore().getProvenValue() == ProvenValue::LOSS && e->getScore().isFinite()) ? 0 : 1;
		return non_losing == 1;
	}
	LowPriorityLock Tree::low_priority_lock() const
	{
		return LowPriorityLock(tree_mutex);
	}
	HighPriorityLock Tree::high_priority_lock() const
	{
		return HighPriorityLock(tree_mutex);
	}
	void Tree::clearNodeCacheStats() noexcept
	{
		stats_baseline = NodeCacheStats();
		stats_baseline = getNodeCacheStats();
	}
	Node Tree::getInfo(const std::vector<Move> &moves) const
	{
		return getInfo(0, moves);
	}
	Node Tree::getInfo(int game, const std::vector<Move> &moves) const
	{
		if (!moves.empty())
			throw std::logic_error("Tree::getInfo() : only the root (an empty move list) can be read from the device");
		std::vector<AgxEdgeView> views;
		const AgxGameInfo info = game_info(bound(), first_game + game, &views);
		std::vector<Edge> edges;
		for (const AgxEdgeView &v : views)
			edges.emplace_back(v);
		return Node(std::move(edges), Value(info.root_win, info.root_draw), Score::from_short(static_cast<uint16_t>(info.root_score)), info.root_visits,
				static_cast<Sign>(info.sign_to_move));
	}
	NodeCacheStats Tree::getNodeCacheStats() const noexcept
	{
		NodeCacheStats result;
		try
		{
			const AgxEngineStats s = bound().getStats();
			// (peaks since agx_engine_begin; after clearNodeCacheStats only what has grown beyond the peaks seen then is reported)
			result.stored_nodes = (s.peak_nodes > stats_baseline.stored_nodes) ? s.peak_nodes : 0;
			result.stored_edges = (s.peak_edges > stats_baseline.stored_edges) ? s.peak_edges : 0;
		} catch (std::exception&)
		{
		}
		return result;
	}

	namespace
	{
		EdgeSelectorConfig unused_final_selector()
		{ // a stand-alone pair never lets the engine pick the move (no advance stage): any valid final selector
			EdgeSelectorConfig c;
			c.policy = "best";
			return c;
		}
	}
	Search::Search(const GameConfig &gameOptions, const SearchConfig &searchOptions) :
			own_pool(std::make_unique<GamePool>(gameOptions, searchOptions, unused_final_selector(), 1, maximum_number_of_simulations, false, "pv", false, 2)),
			pool(*own_pool), group(0), n_groups(2), stream(nullptr), batch_size(searchOptions.max_batch_size)
	{ // the reference's constructor: a one-tree engine (solver table, the TWO task buffers, the arenas of the Tree it will be used with), begun
	  // on the empty board; Tree::setBoard gives it its positions.  Its launches go onto a stream of its own, so that the evaluator's
	  // stream can run the network of one buffer beside the tree work of the other.
		check(agx_stream_create(&own_stream));
		stream = own_stream;
		std::vector<uint16_t> empty_opening(AGX_OPENING_CAP, 0);
		own_pool->begin(empty_opening);
	}
	Search::~Search()
	{
		if (own_stream != nullptr)
		{
			agx_stream_synchronize(own_stream);
			agx_stream_destroy(own_stream);
		}
	}
	Search::Search(GamePool &pool, int group, int n_groups, void *stream) :
			pool(pool), group(group), n_groups(n_groups), stream(stream), batch_size(pool.getBatchSize())
	{
	}
	void Search::bind(Tree &tree)
	{
		if (tree.standalone && tree.pool == nullptr)
		{ // the first meeting of a stand-alone pair: the tree's nodes live in this search's engine from now on
			if (own_pool == nullptr)
				throw std::logic_error("Search : a tree made from a TreeConfig goes with a Search(const GameConfig&, const SearchConfig&)");
			const TreeConfig &want = tree.config, &have = pool.getSearchConfig().tree_config;
			if (want.information_leak_threshold != have.information_leak_threshold)
				throw std::logic_error("Search : the tree's information_leak_threshold differs from the SearchConfig's tree_config");
			tree.pool = &pool;
			tree.stream = stream;
		}
		if (tree.pool != &pool || (own_pool == nullptr && tree.group != group))
			throw std::logic_error("Search : the tree belongs to another search / slice");
	}
	int64_t Search::getMemory() const noexcept
	{
		return 0;
	}
	const SearchConfig& Search::getConfig() const noexcept
	{
		return pool.getSearchConfig();
	}
	AlphaBetaSearch& Search::getSolver() noexcept
	{
		return ab_search;
	}

	/* ---------------------------------------------------------------------------------------------------------------------------------- */
	void SearchTask::set(const matrix<Sign> &base, Sign signToMove)
	{ // SearchTask.cpp:32-50
		if (base.rows() != rows || base.cols() != cols)
			throw std::logic_error("SearchTask::set : the board is " + std::to_string(base.rows()) + "x" + std::to_string(base.cols()) + ", the task " + std::to_string(rows)
					+ "x" + std::to_string(cols));
		board = base;
		sign_to_move = signToMove;
		edges.clear();
		action_scores.fill(Score());
		std::fill(policy.begin(), policy.end(), 0.0f);
		std::fill(action_values.begin(), action_values.end(), Value());
		value = Value();
		score = Score();
		processed_by_network = processed_by_solver = must_defend = statically_solved = recursively_solved = false;
	}
	void SearchTask::addEdge(Move move)
	{ // SearchTask.cpp:62-71
		AgxEdgeView view { };
		view.move = Move(move.row, move.col, sign_to_move).toShort();
		view.score = Score::to_short(action_scores.at(move.row, move.col));
		edges.push_back(Edge(view));
	}

	AlphaBetaSearch::AlphaBetaSearch(const GameConfig &gameConfig) :
			standalone(true), game_config(gameConfig)
	{
		if (gameConfig.rows != gameConfig.cols)
			throw std::logic_error("AlphaBetaSearch : only square boards are supported");
	}
	AlphaBetaSearch::~AlphaBetaSearch()
	{
		if (engine != nullptr)
			agx_engine_destroy(engine);
	}
	void AlphaBetaSearch::require_engine()
	{
		if (!standalone)
			throw std::logic_error("AlphaBetaSearch : the solver of a Search runs inside Search::solve (construct AlphaBetaSearch(GameConfig) for one of its own)");
		if (engine != nullptr)
			return;
		AgxEngineConfig c;
		check(agx_engine_default_config(&c));
		c.rules = static_cast<int>(game_config.rules);
		c.board_size = game_config.rows;
		c.draw_after = game_config.draw_after;
		c.n_games = 1;
		c.max_batch_size = 1;
		c.max_simulations = 1;
		c.node_capacity = 64;
		c.edge_capacity = 1024;
		c.tss_max_positions = max_nodes;
		c.tss_table_entries = static_cast<uint64_t>(table_entries);
		check(agx_engine_create(&c, &engine));
	}
	void AlphaBetaSearch::clear() noexcept
	{ // AlphaBetaSearch.cpp:67-71: an empty table
		clear_requested = true; // (inside a Search: applied by its next cleanup; stand-alone: by the next solve)
	}
	void AlphaBetaSearch::increaseGeneration()
	{ // AlphaBetaSearch.cpp:63-66
		if (!standalone)
			return; // (inside a Search the set-board launch ages the table: Search::setBoard)
		require_engine();
		check(agx_debug_new_generation(engine));
	}
	void AlphaBetaSearch::setNodeLimit(int nodes)
	{ // the budget is a property of the engine: a new limit makes a new one (with an empty table)
		if (nodes < 1)
			throw std::logic_error("AlphaBetaSearch::setNodeLimit : " + std::to_string(nodes));
		if (nodes != max_nodes && engine != nullptr)
		{
			agx_engine_destroy(engine);
			engine = nullptr;
		}
		max_nodes = nodes;
	}
	int AlphaBetaSearch::solve(SearchTask &task)
	{ // AlphaBetaSearch.cpp:77-156
		require_engine();
		const int n = game_config.rows, hw = n * n;
		if (task.getRows() != n || task.getCols() != n)
			throw std::logic_error("AlphaBetaSearch::solve : the task is not of this game's board size");
		if (clear_requested)
		{ // an empty opening list: begin() clears every table of the pool and leaves it idle
			check(agx_engine_destroy(engine));
			engine = nullptr;
			require_engine();
			clear_requested = false;
		}
		std::vector<uint8_t> cells(hw);
		for (int i = 0; i < hw; i++)
			cells[i] = static_cast<uint8_t>(task.getBoard()[i]);
		const int sign = static_cast<int>(task.getSignToMove());
		std::vector<uint16_t> moves(hw), scores(hw);
		int count = 0;
		uint32_t flags = 0;
		uint16_t result = 0;
		check(agx_debug_solve(engine, cells.data(), &sign, 1, task.getFeatures().data(), moves.data(), scores.data(), &count, &flags, &result));
		unsigned long long nodes = 0;
		check(agx_debug_solve_nodes(engine, 1, &nodes));
		for (int i = 0; i < count; i++)
		{
			const Move m(moves[i]);
			const Score sc = Score::from_short(scores[i]);
			task.getActionScores().at(m.row, m.col) = sc;
			if (sc.isProven())
				task.getActionValues()[m.row * n + m.col] = sc.convertToValue();
			task.addEdge(m);
		}
		task.setScore(Score::from_short(result));
		if (task.getScore().isProven())
		{
			task.setValue(task.getScore().convertToValue());
			task.setMovesLeft(static_cast<float>(task.getScore().getDistance()));
		}
		if (flags & 1u)  // TF_MUST_DEFEND
			task.markAsDefensive();
		if (flags & 32u) // TF_RECURSIVELY_SOLVED
			task.maskAsRecursivelySolved();
		if (flags & 16u) // TF_STATICALLY_SOLVED
			task.markAsStaticallySolved();
		task.markAsProcessedBySolver();
		total_positions += static_cast<size_t>(nodes);
		total_calls++;
		return static_cast<int>(nodes);
	}
	void AlphaBetaSearch::print_stats() const
	{
		std::cout << "AlphaBetaSearch : " << total_calls << " calls, " << total_positions << " positions ("
				<< (total_calls > 0 ? static_cast<double>(total_positions) / total_calls : 0.0) << " per call)\n";
	}
	int64_t AlphaBetaSearch::getMemory() const noexcept
	{ // AlphaBetaSearch.cpp:157-160: the table (16 bytes per entry)
		return standalone ? table_entries * 16 : 0;
	}
	void Search::setBoard(const matrix<Sign>&, Sign)
	{ // Search.cpp:112-115: ab_search.increaseGeneration() — the set-board launch Tree::setBoard enqueued ages the solver table (k_set_board)
	}
	void Search::useBuffer(int index)
	{
		if (index != 0 && index != 1)
			throw std::logic_error("Search::useBuffer() : index must be 0 or 1");
		flush_select();
		current_task_buffer = index;
		if (own_pool != nullptr)
			group = index;
	}
	void Search::switchBuffer() noexcept
	{
		try
		{
			useBuffer(1 - current_task_buffer);
		} catch (std::exception&)
		{
		}
	}
	void Search::clearStats() noexcept
	{
		stats = SearchStats();
	}
	SearchStats Search::getStats() const noexcept
	{
		SearchStats result = stats;
		try
		{
			const AgxEngineStats s = pool.getStats();
			result.nb_duplicate_nodes = s.duplicate_selections;
			result.nb_information_leaks = s.information_leaks;
			result.nb_wasted_expansions = s.wasted_expansions;
			result.nb_proven_states = s.proven_edge_visits;
			result.nb_network_evaluations = s.network_evaluations;
			result.nb_node_count = s.evaluated_nodes;
		} catch (std::exception&)
		{
		}
		return result;
	}
	void Search::select(Tree &tree, int maxSimulations)
	{ // a pool slice: the simulation budget is the pool's (SelfplayConfig::constraints.max_simulations, fixed at creation); a stand-alone
	  // pair takes it per call like the reference
		bind(tree);
		if (own_pool != nullptr)
			check(agx_engine_set_max_simulations(pool.handle(), maxSimulations));
		// The launch is enqueued by solve(): on the device a game's wave descends the tree and then solves its leaves in ONE launch
		// (agx_engine_select_solve_group), so a slow descent holds up only its own game.  A caller that never calls solve() gets the
		// stand-alone select launch from the next stage it calls.
		flush_select();
		select_pending = true;
	}
	void Search::flush_select()
	{
		if (select_pending)
		{
			select_pending = false;
			stats.select.startTimer();
			check(agx_engine_select_group(pool.handle(), group, n_groups, stream));
			stats.select.stopTimer();
		}
	}
	void Search::solve(double endTime)
	{
		stats.solve.startTimer();
		if (endTime >= 0.0)
		{ // Search.cpp:159-183 as SearchThread::asynchronous_run calls it: node limit 10 000 and the time until endTime shared out over the leaves
			flush_select();
			const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); // getTime() (utils/misc.hpp:52-55)
			check(agx_engine_solve_timed_group(pool.handle(), group, n_groups, 10000, endTime - now, stream));
			stats.solve.stopTimer();
			scheduled = false;
			return;
		}
		if (select_pending)
		{
			select_pending = false;
			check(agx_engine_select_solve_group(pool.handle(), group, n_groups, stream));
		}
		else
			check(agx_engine_solve_group(pool.handle(), group, n_groups, stream));
		stats.solve.stopTimer();
		scheduled = false;
	}
	void Search::scheduleToNN(NNEvaluator &evaluator)
	{ // Search.cpp:184-199: the solve kernel has compacted the leaves that need the network into the slice's device-side queue
		flush_select();
		stats.schedule.startTimer();
		const int per = (pool.numberOfGames() + n_groups - 1) / n_groups;
		bool &tasks_ready = ready_flags[current_task_buffer];
		tasks_ready = false;
		if (own_pool != nullptr)
			evaluator.addToQueueOverlapped(pool.handle(), current_task_buffer, per * batch_size, stream, &tasks_ready);
		else
			evaluator.addToQueue(pool.handle(), group, n_groups, per * batch_size, stream, &tasks_ready);
		scheduled = true;
		stats.schedule.stopTimer();
	}
	bool Search::areTasksReady() const noexcept
	{
		return ready_flags[current_task_buffer];
	}
	void Search::generateEdges(const Tree&)
	{ // first pass of the expand kernel (UnifiedGenerator::generate per leaf); nothing to enqueue separately
		if (!ready_flags[current_task_buffer])
			throw std::logic_error("Search::generateEdges() : the tasks have not been evaluated yet");
	}
	void Search::expand(Tree &tree)
	{ // one launch: generateEdges for all leaves, then Tree::expand for all, then Tree::backup for all (the order of Search.cpp:206-232)
		if (!ready_flags[current_task_buffer])
			throw std::logic_error("Search::expand() : the tasks have not been evaluated yet");
		stats.expand.startTimer();
		check(agx_engine_expand_group(pool.handle(), group, n_groups, stream));
		tree.summary_valid = false;
		stats.expand.stopTimer();
	}
	void Search::backup(Tree&)
	{ // second half of the launch enqueued by expand()
	}
	void Search::cleanup(Tree &tree)
	{ // Search.cpp:233-242: cancelVirtualLoss of the abandoned tasks of both buffers (a pool slice completes its batch in every step: nothing to cancel)
		bind(tree);
		flush_select();
		tree.summary_valid = false;
		if (own_pool != nullptr)
		{
			check(agx_engine_cancel_pending(pool.handle(), stream));
			ready_flags[0] = ready_flags[1] = true;
		}
		if (ab_search.clear_requested && own_pool != nullptr)
		{ // getSolver().clear() at the start of a game (EvaluationGame.cpp:81-82): an empty table — and an empty tree, which is what the cache
		  // cleanup of the new game's first setBoard would leave of the old game's states anyway
			pool.begin(std::vector<uint16_t>(AGX_OPENING_CAP, 0));
			ab_search.clear_requested = false;
		}
	}
	void Search::setBatchSize(int batchSize)
	{ // Search.cpp:252-255.  A stand-alone Search resizes its buffers (SearchThread.cpp:125-126: the batch grows with sqrt(simulations)); the slices
	  // of a generator thread's pool share one engine and keep the size it was created with (GameGenerator.cpp:31 sets exactly that)
		if (own_pool != nullptr)
		{
			if (batchSize < 1 || batchSize > batch_size)
				throw std::logic_error("Search::setBatchSize() : " + std::to_string(batchSize) + " outside [1, max_batch_size = " + std::to_string(batch_size) + "]");
			check(agx_engine_set_batch_size(pool.handle(), batchSize));
		}
		else if (batchSize != batch_size)
			throw std::logic_error("Search::setBatchSize() : the pool was created with max_batch_size " + std::to_string(batch_size));
	}
	int Search::getBatchSize() const noexcept
	{
		return batch_size;
	}

	/* ------------------------------------------------ selectors ------------------------------------------------ */
	namespace
	{
		/* the search selector: its select() happens inside the device's select stage (dev_mcts.hpp:select_edge); the object only carries the
		 * configuration from EdgeSelector::create to Tree::setEdgeSelector */
		class DeviceSelector: public EdgeSelector
		{
				EdgeSelectorConfig config;
			public:
				explicit DeviceSelector(const EdgeSelectorConfig &cfg) :
						config(cfg)
				{
				}
				std::unique_ptr<EdgeSelector> clone() const override { return std::make_unique<DeviceSelector>(config); }
				const Edge* select(const Node*) noexcept override { return nullptr; }
				const EdgeSelectorConfig& getConfig() const noexcept override { return config; }
		};
		/* the final-move selectors on an owning copy of the root (EdgeSelector.cpp:476-536 under find_best_edge_impl :562-586): one rating
		 * per edge, the first strictly greater one wins */
		class FinalSelector: public EdgeSelector
		{
				EdgeSelectorConfig config;
				int kind;
			public:
				FinalSelector(const EdgeSelectorConfig &cfg, int kind) :
						config(cfg), kind(kind)
				{
				}
				std::unique_ptr<EdgeSelector> clone() const override { return std::make_unique<FinalSelector>(config, kind); }
				const EdgeSelectorConfig& getConfig() const noexcept override { return config; }
				float rate(const Edge &e, int parent_visits) const noexcept
				{
					const ProvenValue pv = e.getScore().getProvenValue();
					const float distance = static_cast<float>(e.getScore().getDistance());
					switch (kind)
					{
						case 0: // 'best' (:515-536): proven results first, else visits + Q * N + a little of the prior
							if (pv == ProvenValue::LOSS)
								return -1.0e8f + distance;
							if (pv == ProvenValue::WIN)
								return +1.0e8f - distance;
							return e.getVisits() + e.getValue().getExpectation() * parent_visits + 0.001f * e.getPolicyPrior();
						case 1: // 'max_visit' (:501-507)
							return static_cast<float>(e.getVisits());
						case 2: // 'min_visit' (:508-514)
							return -static_cast<float>(e.getVisits());
						case 3: // 'max_value' (:476-493)
							if (pv == ProvenValue::LOSS)
								return -1000.0f + distance;
							if (pv == ProvenValue::WIN)
								return +1000.0f - distance;
							return (pv == ProvenValue::DRAW) ? Value(0.0f, 1.0f).getExpectation() : e.getValue().getExpectation();
						default: // 'max_policy' (:494-500)
							return e.getPolicyPrior();
					}
				}
				const Edge* select(const Node *node) noexcept override
				{
					const Edge *best = nullptr;
					float best_rating = std::numeric_limits<float>::lowest();
					for (const Edge *e = node->begin(); e < node->end(); e++)
					{
						const float r = rate(*e, node->getVisits());
						if (r > best_rating)
						{
							best_rating = r;
							best = e;
						}
					}
					return best;
				}
		};
	}
	std::unique_ptr<EdgeSelector> EdgeSelector::create(const EdgeSelectorConfig &config)
	{ // EdgeSelector.cpp:680-711
		if (config.policy == "puct")
			return std::make_unique<DeviceSelector>(config);
		const int kind = selector_id(config.policy); // throws for unknown names
		if (kind == 5)
			throw std::logic_error("EdgeSelector::create() : 'lcb' is provided as the pool's own final selector (on the device), not on root copies");
		return std::make_unique<FinalSelector>(config, kind);
	}

	/* ------------------------------------------------ dataset ------------------------------------------------ */
	std::string GameDataBufferStats::toString() const
	{ // GameDataBuffer.cpp:33-44
		std::string result;
		result += "----GameBufferStats----\n";
		result += "games   = " + std::to_string(games) + '\n';
		result += "samples = " + std::to_string(samples) + '\n';
		result += "cross   = " + std::to_string(cross_win) + '\n';
		result += "draws   = " + std::to_string(draws) + '\n';
		result += "circle  = " + std::to_string(circle_win) + '\n';
		result += "avg len = " + std::to_string(static_cast<float>(game_length) / std::max(1, games)) + '\n';
		return result;
	}
	GameDataBuffer::GameDataBuffer(GameConfig cfg) :
			game_config(cfg)
	{
		check(agx_game_buffer_create(static_cast<int>(cfg.rules), cfg.rows, cfg.cols, cfg.draw_after, &buffer));
	}
	GameDataBuffer::~GameDataBuffer()
	{
		agx_game_buffer_destroy(buffer);
	}
	const GameConfig& GameDataBuffer::getConfig() const noexcept
	{
		return game_config;
	}
	void GameDataBuffer::clear() noexcept
	{
		agx_game_buffer_clear(buffer);
	}
	int GameDataBuffer::numberOfGames() const noexcept
	{
		return getStats().games;
	}
	int GameDataBuffer::numberOfSamples() const noexcept
	{
		return getStats().samples;
	}
	std::vector<uint8_t> GameDataBuffer::getGameData(int index) const
	{
		size_t size = 0;
		check(agx_game_buffer_game(buffer, index, nullptr, 0, &size));
		std::vector<uint8_t> result(size);
		check(agx_game_buffer_game(buffer, index, result.data(), result.size(), &size));
		return result;
	}
	void GameDataBuffer::save(const std::string &path) const
	{
		check(agx_game_buffer_save(buffer, path.c_str(), 1));
	}
	void GameDataBuffer::load(const std::string &path)
	{
		check(agx_game_buffer_load(buffer, path.c_str()));
	}
	GameDataBufferStats GameDataBuffer::getStats() const noexcept
	{
		AgxGameBufferStats s;
		GameDataBufferStats result;
		if (agx_game_buffer_stats(buffer, &s) == AGX_OK)
		{
			result.games = s.games;
			result.samples = s.samples;
			result.cross_win = s.cross_win;
			result.draws = s.draws;
			result.circle_win = s.circle_win;
			result.game_length = s.game_length;
		}
		return result;
	}

	/* ------------------------------------------------ GameGenerator ------------------------------------------------ */
	GameGenerator::GameGenerator(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, GeneratorManager &manager, NNEvaluator &evaluator) :
			manager(manager), nn_evaluator(evaluator), game_config(gameOptions), selfplay_config(selfplayOptions)
	{ // GameGenerator.cpp:36-45: the pool of one game is created by the first generate(), when the evaluator's network (its outputs) is known
		static std::atomic<uint32_t> generator_counter { 0 };
		opening_seed = 7919u * (1u + generator_counter.fetch_add(1u));
	}
	GameGenerator::GameGenerator(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, GeneratorManager &manager, NNEvaluator &evaluator, GamePool &pool,
			int group, int n_groups, void *stream) :
			manager(manager), nn_evaluator(evaluator), game_config(gameOptions), pool(&pool), tree(std::make_unique<Tree>(pool, group, n_groups, stream)),
			search(std::make_unique<Search>(pool, group, n_groups, stream)), selfplay_config(selfplayOptions), group(group), n_groups(n_groups), stream(stream)
	{
	}
	void GameGenerator::start_own_pool()
	{
		own_pool = std::make_unique<GamePool>(game_config, selfplay_config.search_config, selfplay_config.final_selector, 1,
				selfplay_config.constraints.max_simulations, selfplay_config.use_symmetries, nn_evaluator.get_network().getOutputConfig());
		pool = own_pool.get();
		tree = std::make_unique<Tree>(*pool, 0, 1, nullptr);
		search = std::make_unique<Search>(*pool, 0, 1, nullptr);
		own_openings = 0;
		serve_own_pool();
	}
	void GameGenerator::serve_own_pool()
	{ // what the generator thread does for a shared pool: finished games to the manager's buffer, the next openings into the list
	  // (prepareOpening, utils/misc.cpp:142-170; the network-balanced OpeningGenerator is the generator thread's, GeneratorThread::run)
		const bool first = (own_openings == 0);
		if (!first)
			manager.addToBuffer(pool->handle());
		const int taken = first ? 0 : pool->getStats().openings_taken;
		if (first || taken + 1 > own_openings)
		{
			const int count = 8;
			std::vector<uint16_t> list(static_cast<size_t>(count) * AGX_OPENING_CAP, 0);
			for (int i = 0; i < count && selfplay_config.use_opening; i++)
				check(agx_make_opening(static_cast<int>(game_config.rules), game_config.rows, opening_seed++, list.data() + static_cast<size_t>(i) * AGX_OPENING_CAP));
			if (first)
				pool->begin(list);
			else
				pool->addOpenings(list);
			own_openings += count;
		}
	}
	void GameGenerator::clearStats()
	{
		if (search != nullptr)
			search->clearStats();
	}
	NodeCacheStats GameGenerator::getCacheStats() const noexcept
	{
		return (tree != nullptr) ? tree->getNodeCacheStats() : NodeCacheStats();
	}
	SearchStats GameGenerator::getSearchStats() const noexcept
	{
		return (search != nullptr) ? search->getStats() : SearchStats();
	}
	GameGenerator::~GameGenerator()
	{
		for (void *e : pace_events)
			agx_event_destroy(e);
	}
	GameGenerator::Status GameGenerator::generate()
	{ // GameGenerator.cpp:46-121 for every game of the slice at once
		if (pool == nullptr)
			start_own_pool();
		if (state == GAME_NOT_STARTED || state == PREPARE_OPENING)
		{ // beginGame / loadOpening / prepare_search happen on the device when a game takes its opening (k_begin, k_restart); the thread
		  // keeps the opening list ahead of the games (GeneratorThread::run)
			state = GAMEPLAY_SELECT_SOLVE_EVALUATE;
			prepare_search();
		}
		if (state == GAMEPLAY_SELECT_SOLVE_EVALUATE)
		{
			search->select(*tree, selfplay_config.constraints.max_simulations);
			search->solve();
			search->scheduleToNN(nn_evaluator);
			state = GAMEPLAY_EXPAND_AND_BACKUP;
			return GameGenerator::OK;
		}
		if (state == GAMEPLAY_EXPAND_AND_BACKUP)
		{
			if (!search->areTasksReady())
				return GameGenerator::TASKS_NOT_READY;
			search->generateEdges(*tree);
			search->expand(*tree);
			search->backup(*tree);
			// get_simulations_for_move, the move rule, make_move, the end-of-game hand-over and prepare_search (GameGenerator.cpp:97-118) are
			// decided per game on the device; finished games reach manager.addToBuffer through GeneratorThread::collectGames
			make_move();
			state = GAMEPLAY_SELECT_SOLVE_EVALUATE;
			{ // host pacing: nothing here waits for the device, so the thread would enqueue until the launch queue is full and spin there (a CPU
			  // per generator thread); it stays two steps ahead of its stream and sleeps on a blocking event instead
				constexpr uint64_t AHEAD = 2;
				if (pace_events.empty())
				{
					pace_events.assign(AHEAD + 1, nullptr);
					for (void *&e : pace_events)
						check(agx_event_create_blocking(&e));
				}
				check(agx_event_record(pace_events[steps % pace_events.size()], stream));
				if (steps >= AHEAD)
					check(agx_event_synchronize(pace_events[(steps - AHEAD) % pace_events.size()]));
			}
			steps++;
			if (own_pool != nullptr && steps % 64 == 0)
				serve_own_pool();
		}
		return GameGenerator::OK;
	}
	void GameGenerator::save(std::vector<uint8_t> &binary_data)
	{ // GameGenerator.cpp:122-130 for every game of the slice that is in flight
		const size_t count_at = binary_data.size();
		uint32_t count = 0;
		binary_data.insert(binary_data.end(), sizeof(count), 0);
		if (pool != nullptr)
		{
			int in_flight = 0;
			check(agx_engine_save_games(pool->handle(), nullptr, 0, &in_flight));
			std::vector<AgxSavedGame> games(std::max(in_flight, 1));
			check(agx_engine_save_games(pool->handle(), games.data(), static_cast<int>(games.size()), &in_flight));
			const int per = (pool->numberOfGames() + n_groups - 1) / n_groups;
			for (int i = 0; i < in_flight; i++)
			{
				const AgxSavedGame &g = games[i];
				if (g.game_slot / per != group)
					continue;
				size_t bytes = 0;
				check(agx_game_buffer_take_pending(manager.getGameBuffer().handle(), pool->handle(), g.game_slot, g.game_index, nullptr, 0, &bytes));
				std::vector<uint8_t> samples(std::max<size_t>(bytes, 1));
				check(agx_game_buffer_take_pending(manager.getGameBuffer().handle(), pool->handle(), g.game_slot, g.game_index, samples.data(), samples.size(), &bytes));
				const uint8_t *raw = reinterpret_cast<const uint8_t*>(&g);
				binary_data.insert(binary_data.end(), raw, raw + sizeof(AgxSavedGame));
				const uint64_t n = bytes;
				binary_data.insert(binary_data.end(), reinterpret_cast<const uint8_t*>(&n), reinterpret_cast<const uint8_t*>(&n) + sizeof(n));
				binary_data.insert(binary_data.end(), samples.begin(), samples.begin() + bytes);
				count++;
			}
		}
		std::memcpy(binary_data.data() + count_at, &count, sizeof(count));
	}
	size_t GameGenerator::load(const std::vector<uint8_t> &binary_data, size_t offset)
	{ // GameGenerator.cpp:131-141: the Game and its samples come back, the search starts again on an empty tree (prepare_search)
		auto need = [&](size_t n)
		{
			if (offset > binary_data.size() || n > binary_data.size() - offset) // (subtraction form: a length read from the file must not wrap the test)
				throw std::runtime_error("GameGenerator::load() : the saved state is truncated");
		};
		need(sizeof(uint32_t));
		uint32_t count = 0;
		std::memcpy(&count, binary_data.data() + offset, sizeof(count));
		offset += sizeof(count);
		if (pool == nullptr && count > 0)
			start_own_pool();
		const int per = (pool != nullptr) ? (pool->numberOfGames() + n_groups - 1) / n_groups : 1;
		for (uint32_t i = 0; i < count; i++)
		{
			need(sizeof(AgxSavedGame) + sizeof(uint64_t));
			AgxSavedGame g;
			std::memcpy(&g, binary_data.data() + offset, sizeof(g));
			offset += sizeof(g);
			uint64_t n = 0;
			std::memcpy(&n, binary_data.data() + offset, sizeof(n));
			offset += sizeof(n);
			need(n);
			if (g.game_slot >= 0 && g.game_slot < pool->numberOfGames() && g.game_slot / per == group)
			{
				check(agx_engine_restore_game(pool->handle(), &g, stream));
				check(agx_game_buffer_restore_pending(manager.getGameBuffer().handle(), pool->handle(), g.game_slot, 0, binary_data.data() + offset, n));
			}
			offset += n;
		}
		if (state == GAMEPLAY_EXPAND_AND_BACKUP)
			state = GAMEPLAY_SELECT_SOLVE_EVALUATE;
		return offset;
	}
	void GameGenerator::make_move()
	{ // GameGenerator.cpp:145-173 for the games whose root has its visits: final selector, sample, Game::makeMove
		check(agx_engine_advance_group(pool->handle(), group, n_groups, stream));
	}
	void GameGenerator::prepare_search()
	{ // GameGenerator.cpp:174-185: cleanup + Tree::setBoard + fresh selector / generator — part of the advance kernel on the device
		search->cleanup(*tree);
	}

	/* ------------------------------------------------ GeneratorThread ------------------------------------------------ */
	GeneratorThread::GeneratorThread(GeneratorManager &manager, const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, int index) :
			is_running(true), manager(manager), nn_evaluator(selfplayOptions.device_config.at(index)), game_config(gameOptions), selfplay_config(selfplayOptions),
			index(index)
	{
		nn_evaluator.useSymmetries(selfplayOptions.use_symmetries);
	}
	GeneratorThread::~GeneratorThread()
	{
		if (generator_future.valid())
			generator_future.wait();
	}
	void GeneratorThread::start()
	{
		is_running.store(true);
		generator_future = std::async(std::launch::async, [this]()
		{
			try
			{
				this->run();
			}
			catch(std::exception &e)
			{
				std::cout << "GeneratorThread::start() threw " << e.what() << '\n';
				exit(-1);
			}
		});
	}
	void GeneratorThread::stop()
	{
		is_running.store(false);
		if (generator_future.valid())
			generator_future.wait();
	}
	bool GeneratorThread::isFinished() const noexcept
	{
		if (generator_future.valid())
			return generator_future.wait_for(std::chrono::milliseconds(0)) == std::future_status::ready;
		return true;
	}
	void GeneratorThread::clearStats() noexcept
	{
		nn_evaluator.clearStats();
	}
	void GeneratorThread::setWorkingDirectory(const std::string &path)
	{
		working_directory = path;
	}
	NNEvaluatorStats GeneratorThread::getEvaluatorStats() const noexcept
	{
		std::lock_guard<std::mutex> lock(stats_mutex);
		NNEvaluatorStats result = nn_evaluator.getStats();
		result.batch_sizes += last_search_stats.nb_network_evaluations;
		return result;
	}
	NodeCacheStats GeneratorThread::getCacheStats() const noexcept
	{
		std::lock_guard<std::mutex> lock(stats_mutex);
		return last_cache_stats;
	}
	SearchStats GeneratorThread::getSearchStats() const noexcept
	{
		std::lock_guard<std::mutex> lock(stats_mutex);
		return last_search_stats;
	}
	void GeneratorThread::collectGames()
	{
		manager.addToBuffer(pool->handle());
		const SearchStats s = generators.front()->getSearchStats();
		const NodeCacheStats c = generators.front()->getCacheStats();
		std::lock_guard<std::mutex> lock(stats_mutex);
		last_search_stats = s;
		last_cache_stats = c;
	}
	static std::vector<uint8_t> saved_games_header(uint32_t generators)
	{ // "AGXS", u32 version, u32 number of generators
		std::vector<uint8_t> out({ 'A', 'G', 'X', 'S', 1, 0, 0, 0 });
		out.insert(out.end(), reinterpret_cast<const uint8_t*>(&generators), reinterpret_cast<const uint8_t*>(&generators) + 4);
		return out;
	}
	void GeneratorThread::saveGames(const std::string &path) const
	{ // GeneratorManager.cpp:98-110.  File: "AGXS", u32 version, u32 generators, then every generator's GameGenerator::save bytes.  A thread that
	  // has no games in flight — never started, or its run() ended early — writes a valid state of zero generators (the reference writes an
	  // empty Json then), which loadGames accepts: a 0-byte file would make the next start's loadState() throw until saved_state/ is removed by hand
		if (!isFinished())
			throw std::logic_error("GeneratorThread::saveGames() : cannot save while the generator is running");
		std::ofstream out(path, std::ofstream::out | std::ofstream::binary);
		if (!out.good())
			throw std::runtime_error("GeneratorThread::saveGames() : cannot open '" + path + "'");
		const std::vector<uint8_t> empty = saved_games_header(0);
		const std::vector<uint8_t> &bytes = saved_games.empty() ? empty : saved_games;
		out.write(reinterpret_cast<const char*>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
	}
	void GeneratorThread::loadGames(const std::string &path)
	{ // GeneratorManager.cpp:111-122; the games continue when the thread is started next (run() hands them to its generators)
		if (!isFinished())
			throw std::logic_error("GeneratorThread::loadGames() : cannot load while the generator is running");
		std::ifstream in(path, std::ifstream::in | std::ifstream::binary);
		if (!in.good())
			throw std::runtime_error("GeneratorThread::loadGames() : cannot open '" + path + "'");
		saved_games.assign((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
		if (saved_games.size() < 12 || std::memcmp(saved_games.data(), "AGXS", 4) != 0)
		{
			saved_games.clear();
			throw std::runtime_error("GeneratorThread::loadGames() : '" + path + "' is not a saved generator state");
		}
	}
	void GeneratorThread::setup()
	{
		const DeviceConfig &device = selfplay_config.device_config.at(index);
		check(agx_set_device(device.device.index()));
		nn_evaluator.loadGraph(manager.getNetworkLoader());
		const int games = selfplay_config.games_per_thread;
		pool = std::make_unique<GamePool>(game_config, selfplay_config.search_config, selfplay_config.final_selector, games,
				selfplay_config.constraints.max_simulations, selfplay_config.use_symmetries, nn_evaluator.get_network().getOutputConfig());
		// slices: as many network launches as the queue capacity (games x max_batch_size) needs at DeviceConfig::batch_size positions each
		const int slots = games * selfplay_config.search_config.max_batch_size;
		int n_groups = (device.batch_size > 0) ? (slots + device.batch_size - 1) / device.batch_size : 1;
		n_groups = std::max(1, std::min(std::min(16, games), n_groups));
		// Slices own disjoint blocks of the chip's compute units (CU-masked streams, agx.h): out of phase with each other, so the
		// power-limited network launches never cover the whole chip at once and no slice waits for another's stragglers.
		int cus = 0;
		check(agx_device_cu_count(&cus));
		bool partition = n_groups > 1 && cus >= n_groups && games % n_groups == 0;
		std::vector<void*> masked;
		for (int g = 0; partition && g < n_groups; g++)
		{
			std::vector<uint32_t> mask((cus + 31) / 32, 0u);
			for (int c = g * (cus / n_groups); c < (g + 1) * (cus / n_groups); c++)
				mask[c / 32] |= 1u << (c % 32);
			void *s = nullptr;
			// (instance = this thread's index: two generator threads on one device — several device_config entries may name it — must not share a queue)
			if (agx_stream_create_with_cu_mask_instance(&s, mask.data(), static_cast<int>(mask.size()), index) == AGX_OK)
				masked.push_back(s);
			else
				partition = false; // no CU masks on this device / runtime: plain streams (the slices then only pipeline host work)
		}
		if (partition)
			check(agx_net_set_launch_width(nn_evaluator.get_network().handle(), cus / n_groups));
		for (int g = 0; g < n_groups; g++)
		{
			void *s = nullptr;
			if (partition)
				s = masked[g];
			else
				check(agx_stream_create(&s));
			streams.push_back(s);
			generators.push_back(std::make_unique<GameGenerator>(game_config, selfplay_config, manager, nn_evaluator, *pool, g, n_groups, s));
		}
	}
	void GeneratorThread::teardown()
	{
		generators.clear();
		for (void *s : streams)
			agx_stream_destroy(s);
		streams.clear();
		pool.reset();
	}
	void GeneratorThread::run()
	{ // GeneratorManager.cpp:124-141
		setup();
		const int games = selfplay_config.games_per_thread;
		const int n = game_config.rows;
		uint32_t next_seed = static_cast<uint32_t>(index) * 1000003u;
		auto make_openings = [&](int count)
		{ // OpeningGenerator (selfplay/OpeningGenerator.cpp:21-78): random openings that the solver cannot prove and the network finds balanced
			std::vector<uint16_t> out(static_cast<size_t>(count) * AGX_OPENING_CAP, 0);
			if (!selfplay_config.use_opening)
				return out; // empty boards
			AgxEngineConfig c;
			check(agx_engine_default_config(&c));
			c.rules = static_cast<int>(game_config.rules);
			c.board_size = n;
			c.draw_after = game_config.draw_after;
			c.n_games = 64; // candidates evaluated together
			c.max_batch_size = 1;
			c.tss_max_positions = 1000; // OpeningGenerator.cpp:61
			c.tss_table_entries = 1u << 18;
			c.node_capacity = 16;
			c.edge_capacity = 1024;
			c.action_values = (nn_evaluator.get_network().getOutputConfig() == "pvq") ? 1 : 0;
			AgxEngine *helper = nullptr;
			check(agx_engine_create(&c, &helper));
			const int st = agx_engine_generate_openings(helper, nn_evaluator.get_network().handle(), count, next_seed, out.data(), nullptr);
			agx_engine_destroy(helper);
			check(st);
			next_seed += 16u * static_cast<uint32_t>(count);
			return out;
		};
		int n_openings = games + games / 2;
		pool->begin(make_openings(n_openings));
		if (!saved_games.empty())
		{ // loadGames: every saved game goes to the generator whose slice holds its slot (a file written with another slicing is read by all of them)
			uint32_t saved_generators = 0;
			std::memcpy(&saved_generators, saved_games.data() + 8, 4);
			size_t offset = 12;
			for (uint32_t k = 0; k < saved_generators; k++)
			{
				const size_t begin = offset;
				for (size_t i = 0; i < generators.size(); i++)
					offset = generators[i]->load(saved_games, begin);
			}
			saved_games.clear();
		}

		uint64_t iterations = 0;
		// The slices' streams run their queues independently, so left alone every slice starts its first search launch at the same moment and
		// the slices stay in step: all their network launches at once — the power-limited case the slicing exists to avoid.  The odd slices
		// therefore begin when slice 0's first search launch is over (an event on its stream): two network launches at a time, not all.
		struct EventGuard
		{ // (destroyed on every way out of run(), a stage that throws included: the exception ends the process, GeneratorThread::start)
				void *event = nullptr;
				~EventGuard()
				{
					if (event != nullptr)
						agx_event_destroy(event);
				}
		} first_search_done;
		while (is_running.load() and not manager.hasEnoughGames())
		{
			for (size_t i = 0; i < generators.size(); i++)
			{
				if (iterations == 0 && i % 2 == 1 && first_search_done.event != nullptr)
					check(agx_stream_wait_event(streams[i], first_search_done.event));
				const GameGenerator::Status status = generators[i]->generate();
				if (iterations == 0 && i == 0 && generators.size() > 1)
				{
					check(agx_event_create(&first_search_done.event));
					check(agx_event_record(first_search_done.event, streams[0]));
				}
				if (nn_evaluator.isQueueFull() or status == GameGenerator::TASKS_NOT_READY)
				{
					nn_evaluator.asyncEvaluateGraphJoin();
					nn_evaluator.asyncEvaluateGraphLaunch();
				}
			}
			if (++iterations % 256 == 0)
			{ // hand the finished games over and keep the opening list ahead of the games
				collectGames();
				const AgxEngineStats st = pool->getStats();
				if (st.first_error != 0)
					throw std::runtime_error("the device engine stopped a game with error " + std::to_string(st.first_error));
				if (st.openings_taken + games > n_openings)
				{ // slot s plays openings s, s + games, s + 2 games, ...: keep a whole round beyond the furthest slot in the list
					pool->addOpenings(make_openings(games));
					n_openings += games;
				}
			}
		}
		nn_evaluator.asyncEvaluateGraphJoin();
		check(agx_device_synchronize());
		collectGames();
		park_games_in_flight();
		nn_evaluator.unloadGraph();
		teardown();
	}
	void GeneratorThread::park_games_in_flight()
	{ // what saveGames will write: the games still in flight, slice by slice (their samples leave the manager's buffer with them)
		saved_games = saved_games_header(static_cast<uint32_t>(generators.size()));
		for (size_t i = 0; i < generators.size(); i++)
			generators[i]->save(saved_games);
		agx_game_buffer_forget_engine(manager.getGameBuffer().handle(), pool->handle());
	}

	/* ------------------------------------------------ signals (utils/os_utils.cpp:20-205) ------------------------------------------------ */
	namespace
	{
		volatile std::sig_atomic_t captured_signals[6] = { 0, 0, 0, 0, 0, 0 };
		int signal_number(SignalType type) noexcept
		{
			switch (type)
			{
				case SignalType::INT: return SIGINT;
				case SignalType::ILL: return SIGILL;
				case SignalType::ABRT: return SIGABRT;
				case SignalType::FPE: return SIGFPE;
				case SignalType::SEGV: return SIGSEGV;
				default: return SIGTERM;
			}
		}
		template<int K>
		void capture(int)
		{
			captured_signals[K] = 1;
		}
	}
	void setupSignalHandler(SignalType type, SignalHandlerMode mode)
	{
		typedef void (*Handler)(int);
		static const Handler custom[6] = { capture<0>, capture<1>, capture<2>, capture<3>, capture<4>, capture<5> };
		switch (mode)
		{
			case SignalHandlerMode::DEFAULT_HANDLER:
				std::signal(signal_number(type), SIG_DFL);
				break;
			case SignalHandlerMode::IGNORE_SIGNAL:
				std::signal(signal_number(type), SIG_IGN);
				break;
			case SignalHandlerMode::CUSTOM_HANDLER:
				std::signal(signal_number(type), custom[static_cast<int>(type)]);
				break;
		}
	}
	bool hasCapturedSignal(SignalType type) noexcept
	{
		return captured_signals[static_cast<int>(type)] != 0;
	}

	/* ------------------------------------------------ GeneratorManager ------------------------------------------------ */
	GeneratorManager::GeneratorManager(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions) :
			generators(selfplayOptions.device_config.size()), game_buffer(gameOptions)
	{
		for (size_t i = 0; i < generators.size(); i++)
			generators[i] = std::make_unique<GeneratorThread>(*this, gameOptions, selfplayOptions, static_cast<int>(i));
	}
	void GeneratorManager::setWorkingDirectory(const std::string &path)
	{
		working_directory = path;
		for (size_t i = 0; i < generators.size(); i++)
			generators[i]->setWorkingDirectory(path);
	}
	int GeneratorManager::addToBuffer(AgxEngine *engine)
	{ // GeneratorManager.cpp:160-164
		std::lock_guard<std::mutex> lock(buffer_mutex);
		int added = 0;
		check(agx_game_buffer_collect(game_buffer.handle(), engine, &added));
		return added;
	}
	const GameDataBuffer& GeneratorManager::getGameBuffer() const noexcept
	{
		return game_buffer;
	}
	GameDataBuffer& GeneratorManager::getGameBuffer() noexcept
	{
		return game_buffer;
	}
	const NetworkLoader& GeneratorManager::getNetworkLoader() const noexcept
	{
		return network_loader;
	}
	bool GeneratorManager::hasEnoughGames() const noexcept
	{
		std::lock_guard<std::mutex> lock(buffer_mutex);
		return game_buffer.numberOfGames() >= games_to_generate;
	}
	void GeneratorManager::generate(const NetworkLoader &loader, int numberOfGames)
	{ // GeneratorManager.cpp:182-218: start every device's thread, wait until all of them have seen hasEnoughGames(); the statistics every 60 s
	  // (and once at the end of a shorter call); a captured SIGINT stops the threads and returns, so that the caller gets to saveState
	  // (TrainingManager.cpp:202-209).  The polling interval is 20 ms instead of 1 s — a pool finishes games by the hundred per second.
		games_to_generate = numberOfGames;
		network_loader = loader;
		for (auto &thread : generators)
		{
			thread->clearStats();
			thread->start();
		}
		auto all_finished = [this]()
		{
			return std::all_of(generators.begin(), generators.end(), [](const std::unique_ptr<GeneratorThread> &t) { return t->isFinished(); });
		};
		const auto begun = std::chrono::steady_clock::now();
		long long minutes_reported = 0;
		while (!all_finished())
		{
			std::this_thread::sleep_for(std::chrono::milliseconds(20));
			const long long minutes = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - begun).count() / stats_period_seconds;
			if (minutes > minutes_reported)
			{
				minutes_reported = minutes;
				printStats();
			}
			if (hasCapturedSignal(SignalType::INT))
			{
				std::cout << "Caught interruption signal" << std::endl;
				for (auto &thread : generators)
					thread->stop(); // (joins: the thread parks its games in flight on the way out, GeneratorThread::run)
				std::cout << "Generators stopped" << std::endl;
				return;
			}
		}
		if (minutes_reported == 0)
			printStats();
	}
	void GeneratorManager::printStats()
	{ // GeneratorManager.cpp:219-240: progress, the buffer's summary, then the per-thread statistics averaged over the threads
		const int threads = static_cast<int>(generators.size());
		NNEvaluatorStats evaluator_total;
		SearchStats search_total;
		NodeCacheStats cache_total;
		for (const auto &thread : generators)
		{
			evaluator_total += thread->getEvaluatorStats();
			search_total += thread->getSearchStats();
			cache_total += thread->getCacheStats();
		}
		if (threads > 0)
		{
			evaluator_total /= threads;
			search_total /= threads;
			cache_total /= threads;
		}
		std::ostringstream text;
		text << "Played games = " << game_buffer.numberOfGames() << "/" << games_to_generate << '\n' << game_buffer.getStats().toString() << '\n';
		text << evaluator_total.toString() << search_total.toString() << cache_total.toString();
		std::cout << text.str() << std::endl;
	}
	void GeneratorManager::saveState(bool saveBuffer)
	{ // GeneratorManager.cpp:241-262: saved_state/buffer.bin (optional) and one saved_state/thread_<i>.bin per generator thread with its games in flight
		if (working_directory.empty())
			return;
		const std::string path = working_directory + "/saved_state/";
		std::filesystem::create_directories(path);
		if (saveBuffer)
		{
			std::cout << "Saving buffer" << std::endl;
			game_buffer.save(path + "buffer.bin");
		}
		std::cout << "Saving games" << std::endl;
		for (size_t i = 0; i < generators.size(); i++)
		{
			const std::string file = path + "thread_" + std::to_string(i) + ".bin";
			generators[i]->saveGames(file);
			std::cout << "Saved " << file << std::endl;
		}
	}
	void GeneratorManager::loadState()
	{ // GeneratorManager.cpp:263-290: the buffer comes back (and its file is removed, as in the reference), every thread gets its games back
		if (working_directory.empty())
			return;
		const std::string path = working_directory + "/saved_state/";
		if (!std::filesystem::exists(path))
		{
			std::cout << "No saved state was found" << std::endl;
			return;
		}
		if (std::filesystem::exists(path + "buffer.bin"))
		{
			game_buffer.load(path + "buffer.bin");
			std::cout << "Loaded buffer:\n" << game_buffer.getStats().toString() << std::endl;
			std::filesystem::remove(path + "buffer.bin");
		}
		for (size_t i = 0; i < generators.size(); i++)
		{
			const std::string file = path + "thread_" + std::to_string(i) + ".bin";
			if (std::filesystem::exists(file))
			{
				generators[i]->loadGames(file);
				std::cout << "Loaded " << file << std::endl;
			}
		}
	}
} /* namespace ag */
