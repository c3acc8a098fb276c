// rsx_multi_state.hpp: the per-rank state and helpers of rsx_sort_multi (inside rsx.hip's anonymous namespace) -- part of librsx.so's host side; included by rsx.hip at the point where it used to stand (one translation unit:
// the kernels' instantiations are shared).  See rsx.hip for the context type, the error convention and the helpers used here.
#pragma once

// ---- single-process multi-device sort (rsx_sort_multi) -----------------------------------------------------------------
// A "rank" is a (device, stream) pair: its own workspace context, its own host thread while a phase runs.  The streams are
// pooled per (device, slot) so that repeated calls reuse the contexts.
std::mutex g_multi_mu;
std::map<std::pair<int, int>, hipStream_t> g_multi_streams;
// device buffers of a rank, kept per (device, slot) between calls (grown when a larger sort comes, freed by rsx_release)
struct MultiBufs {
	DevBuf shard, part, recv, aux, misc;   // misc: [key bytes][256] u64 histogram + 64 bytes for the flag
};
std::map<std::pair<int, int>, MultiBufs> g_multi_bufs;
std::map<std::pair<int, int>, int> g_peer_enabled;   // (device, peer) -> 1 enabled, 0 not possible

struct MultiRank {
	int dev = 0;
	MultiBufs *bufs = nullptr;
	hipStream_t stream = nullptr;
	size_t first = 0, count = 0;         // this rank's shard of the input
	void *shard = nullptr, *part = nullptr, *recv = nullptr, *aux = nullptr;
	u64 *d_hist = nullptr;
	u32 *d_flag = nullptr;
	std::vector<u64> hist;               // [key bytes][256]
	u32 unsorted = 0;
	size_t n_recv = 0, out_first = 0;    // this rank's range of the result
	int rc = RSX_OK;
	char err[512] = "";
};

u64 host_kdf(const void *p, size_t kb, int dtype, int order)
{
	u64 raw = 0;
	memcpy(&raw, p, kb);
	const u64 ones = kb == 8 ? ~0ull : ((1ull << (8 * kb)) - 1);
	const u64 high = 1ull << (8 * kb - 1);
	const bool is_signed = dtype == RSX_I8 || dtype == RSX_I16 || dtype == RSX_I32 || dtype == RSX_I64;
	const bool is_float = dtype == RSX_F32 || dtype == RSX_F64;
	u64 k = raw;
	if (is_float)
		k ^= (raw & high) ? ones : high;     // radix_sort_basic_kdf.hpp:32-46
	else if (is_signed)
		k ^= high;                           // :26-30
	if (order == RSX_DESCENDING)
		k ^= ones;                           // README.md:564-574
	return k & ones;
}

// digit -> destination rank: contiguous, monotone ranges of ~total/G keys (the midpoint of a digit's run decides)
void choose_splitters_host(const u64 *hist, int G, uint8_t *lut)
{
	long double total = 0;
	for (int d = 0; d < 256; ++d)
		total += (long double)hist[d];
	long double before = 0;
	int prev = 0;
	for (int d = 0; d < 256; ++d) {
		int r = 0;
		if (total > 0 && G > 1) {
			const long double mid = before + (long double)hist[d] / 2;
			r = (int)(mid * G / total);
			r = r < 0 ? 0 : (r > G - 1 ? G - 1 : r);
		}
		if (r < prev)
			r = prev;
		prev = r;
		lut[d] = (uint8_t)r;
		before += (long double)hist[d];
	}
}

// one host thread per rank; the first failure (code + message) becomes the caller's
int multi_phase(std::vector<MultiRank> &ranks, const std::function<int(MultiRank &)> &body)
{
	std::vector<std::thread> threads;
	for (auto &r : ranks)
		threads.emplace_back([&r, &body]() {
			if (hipSetDevice(r.dev) != hipSuccess) {
				r.rc = RSX_EHIP;
				snprintf(r.err, sizeof(r.err), "hipSetDevice(%d) failed", r.dev);
				return;
			}
			r.rc = body(r);
			if (r.rc != RSX_OK)
				snprintf(r.err, sizeof(r.err), "%s", g_err);
		});
	for (auto &t : threads)
		t.join();
	for (auto &r : ranks)
		if (r.rc != RSX_OK)
			return fail(r.rc, "rsx_sort_multi (device %d): %s", r.dev, r.err);
	return RSX_OK;
}

// the ranks' buffers belong to g_multi_bufs; a call only has to be sure that nothing of it is still running
void multi_quiesce(std::vector<MultiRank> &ranks)
{
	for (auto &r : ranks) {
		(void)hipSetDevice(r.dev);
		if (r.stream)
			(void)hipStreamSynchronize(r.stream);
	}
}

// direct peer-to-peer copies dev <- peer over xGMI where the topology allows them (without it hipMemcpyPeerAsync stages
// through the host); enabled once per ordered pair of devices
void enable_peer(int dev, int peer)
{
	if (dev == peer)
		return;
	std::lock_guard<std::mutex> lock(g_mu);
	const auto key = std::make_pair(dev, peer);
	if (g_peer_enabled.count(key))
		return;
	int can = 0;
	int ok = 0;
	if (hipDeviceCanAccessPeer(&can, dev, peer) == hipSuccess && can) {
		const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);   // (for the current device, which is `dev` here)
		ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
	}
	(void)hipGetLastError();
	g_peer_enabled[key] = ok;
}

