/*
 * alphagomoku_agx/networks.hpp — ag::AGNetwork and ag::NetworkLoader over the HIP tower kernels (agx_nn_*), keeping the reference's
 * call surface for the inference side (include/alphagomoku/networks/AGNetwork.hpp:40-112, include/alphagomoku/selfplay/NetworkLoader.hpp:
 * 22-31).  The reference's AGNetwork wraps an ml::Graph of the absent MinML library; here the graph is the fixed ResnetPV / ResnetPVQ
 * tower of src/networks/networks.cpp:71-93,143-168 with BatchNorm already folded (the state after AGNetwork::optimize,
 * src/networks/AGNetwork.cpp:136-160).  Training entry points (train, getLoss, changeLearningRate, init) are not part of the path.
 *
 * Weight files: MinML's serialisation format is not in the reference tree, so a checkpoint cannot be imported; saveToFile / loadFromFile
 * use a plain container: "AGXW", the AgxNetDesc (7 ints), u64 count, the canonical fp32 blob of include/agx.h.
 */
#ifndef ALPHAGOMOKU_AGX_NETWORKS_HPP_
#define ALPHAGOMOKU_AGX_NETWORKS_HPP_

#include "configs.hpp"
#include "../agx.h"

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace ag
{
	struct Value
	{ // search/Value.hpp:26-114
			float win_rate = 0.0f;
			float draw_rate = 0.0f;
			Value() = default;
			Value(float w, float d = 0.0f) :
					win_rate(w), draw_rate(d)
			{
			}
			float loss_rate() const noexcept
			{
				return 1.0f - win_rate - draw_rate;
			}
			float getExpectation() const noexcept
			{
				return win_rate + 0.5f * draw_rate;
			}
			Value getInverted() const noexcept
			{
				return Value(loss_rate(), draw_rate);
			}
	};

	class AGNetwork
	{
			GameConfig game_config;
			AgxNetDesc desc { };
			AgxNet *net = nullptr;
			int batch_size = 0;
			// host staging of the reference's NetworkDataPack + the device tensors behind it
			std::vector<uint32_t> input;
			std::vector<float> policy, value, action_values;
			void *d_input = nullptr, *d_policy = nullptr, *d_value = nullptr, *d_action_values = nullptr;
			int launched = 0;
		public:
			AGNetwork() noexcept = default;
			/* architecture "ResnetPV" / "ResnetPVraw" (outputs "pv"; the raw network reads the 8 low bits of a feature word) or "ResnetPVQ" ("pvq") */
			AGNetwork(const GameConfig &gameOptions, const std::string &architecture, int blocks, int filters);
			AGNetwork(const AGNetwork &other) = delete;
			AGNetwork& operator=(const AGNetwork &other) = delete;
			~AGNetwork();

			std::string getOutputConfig() const;
			std::string name() const;
			size_t numberOfWeights() const;
			void loadWeights(const std::vector<float> &blob); // the canonical blob of include/agx.h

			/* "Can be used to pack the data if the features were already calculated" (AGNetwork.hpp:66-69): one uint32 per cell */
			void packInputData(int index, const uint32_t *features);
			void unpackOutput(int index, std::vector<float> &policy, std::vector<Value> &actionValues, Value &value, float &movesLeft) const;
			void asyncForwardLaunch(int batch_size);
			void asyncForwardJoin();
			void forward(int batch_size);

			void optimize(int level = 1);      // BatchNorm is folded in the blob already
			void convertToHalfFloats();        // the kernels store weights and activations as fp16 and accumulate in fp32
			void saveToFile(const std::string &path) const;
			void loadFromFile(const std::string &path);
			void unloadGraph();
			bool isLoaded() const noexcept;
			void synchronize();
			void moveTo(Device device);
			int getBatchSize() const noexcept;
			void setBatchSize(int batchSize);
			GameConfig getGameConfig() const noexcept;
			AgxNet* handle() const noexcept
			{
				return net;
			}
		private:
			std::vector<float> blob_copy; // kept so that saveToFile / moveTo can re-create the device copy
			void release();
			void create();
	};
	std::unique_ptr<AGNetwork> loadAGNetwork(const std::string &path);

	class NetworkLoader
	{ // selfplay/NetworkLoader.hpp:22-31; several paths = the average of their weights (NetworkLoader.cpp:44-56, SWA)
			std::vector<std::string> paths;
		public:
			NetworkLoader() noexcept = default;
			NetworkLoader(const char *path);
			NetworkLoader(const std::string &path);
			NetworkLoader(const std::vector<std::string> &path);
			std::unique_ptr<AGNetwork> get(bool optimized = true) const;
	};
} /* namespace ag */

#endif
