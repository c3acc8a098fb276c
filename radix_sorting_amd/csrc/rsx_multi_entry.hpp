// rsx_multi_entry.hpp: rsx_sort_multi (inside rsx.hip's extern "C" block) -- part of librsx.so's host side; included by rsx.hip at the point where it used to stand (one translation unit:
// the kernels' instantiations are shared).  See rsx.hip for the context type, the error convention and the helpers used here.
#pragma once

// radix_sort(src, aux, n) on host buffers with the work spread over several devices of ONE process (SURVEY.md 8b item 5;
// the one-process-per-GPU form of the same algorithm is radix_sorting_amd/multi.py).  The front half of rs_sort_main is
// done globally -- column histograms summed over the shards, the ordered-neighbour test across shard boundaries, the
// column probe on src[0] -- so early exit, kept columns and the returned pointer are exactly the reference's; then every
// shard is split by the highest kept byte, the devices pull their digit ranges from each other, sort them and write them
// to their place in the buffer the parity rule names.
int rsx_sort_multi(void *src, void *aux, size_t n, rsx_dtype dtype, rsx_order order, const int *devices, int ndev,
                   void **result, rsx_info *info)
{
	info_clear(info, dtype);
	const size_t kb = dtype_size(dtype);
	if (!kb || !result || ndev < 1 || ndev > 64 || !devices || (n && (!src || !aux)))
		return fail(RSX_EINVAL, "rsx_sort_multi: bad argument");
	if (n < 2) {                             // radix_sort.hpp:100-101
		*result = src;
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	{
		std::lock_guard<std::mutex> lock(g_mu);
		if (probe_devices() <= 0)
			return fail(RSX_ENODEVICE, "no gfx950 (MI355X) device visible to HIP; this library has no CPU path");
	}
	int visible = 0;
	HIP_TRY(hipGetDeviceCount(&visible));
	for (int i = 0; i < ndev; ++i)
		if (devices[i] < 0 || devices[i] >= visible)
			return fail(RSX_EINVAL, "rsx_sort_multi: device %d is not one of the %d visible", devices[i], visible);
	std::lock_guard<std::mutex> multi_lock(g_multi_mu);
	int home = 0;
	HIP_TRY(hipGetDevice(&home));
	const int G = ndev;
	std::vector<MultiRank> ranks(G);
	std::map<int, int> slot;
	for (int r = 0; r < G; ++r) {
		MultiRank &k = ranks[r];
		k.dev = devices[r];
		const auto key = std::make_pair(k.dev, slot[k.dev]++);
		auto it = g_multi_streams.find(key);
		if (it == g_multi_streams.end()) {
			hipStream_t st = nullptr;
			HIP_TRY(hipSetDevice(k.dev));
			HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
			it = g_multi_streams.emplace(key, st).first;
		}
		k.stream = it->second;
		k.bufs = &g_multi_bufs[key];
		k.first = (size_t)((unsigned __int128)n * r / G);
		k.count = (size_t)((unsigned __int128)n * (r + 1) / G) - k.first;
		k.hist.assign(kb * 256, 0);
	}
	struct Cleanup {
		std::vector<MultiRank> &ranks;
		int home;
		~Cleanup()
		{
			multi_quiesce(ranks);
			(void)hipSetDevice(home);
		}
	} cleanup{ranks, home};
	const char *hsrc = (const char *)src;

	// ---- radix_sort.hpp:47-58 per shard: upload, histogram of every column, ordered-neighbour test
	RSX_TRY(multi_phase(ranks, [&](MultiRank &k) -> int {
		if (k.count == 0)
			return RSX_OK;
		RSX_TRY(k.bufs->shard.ensure(k.count * kb));
		RSX_TRY(k.bufs->part.ensure(k.count * kb));
		RSX_TRY(k.bufs->misc.ensure(kb * 256 * sizeof(u64) + 64));
		k.shard = k.bufs->shard.p;
		k.part = k.bufs->part.p;
		k.d_hist = (u64 *)k.bufs->misc.p;
		k.d_flag = (u32 *)((char *)k.bufs->misc.p + kb * 256 * sizeof(u64));
		HIP_TRY(hipMemcpyAsync(k.shard, hsrc + k.first * kb, k.count * kb, hipMemcpyHostToDevice, k.stream));
		RSX_TRY(rsx_histogram_device(k.shard, k.count, dtype, order, (uint64_t *)k.d_hist, (uint32_t *)k.d_flag, k.stream));
		HIP_TRY(hipMemcpyAsync(k.hist.data(), k.d_hist, kb * 256 * sizeof(u64), hipMemcpyDeviceToHost, k.stream));
		HIP_TRY(hipMemcpyAsync(&k.unsorted, k.d_flag, sizeof(u32), hipMemcpyDeviceToHost, k.stream));
		HIP_TRY(hipStreamSynchronize(k.stream));
		return RSX_OK;
	}));
	std::vector<u64> ghist(kb * 256, 0);
	bool sorted = true;
	for (int r = 0; r < G; ++r) {
		for (size_t i = 0; i < kb * 256; ++i)
			ghist[i] += ranks[r].hist[i];
		sorted = sorted && ranks[r].unsorted == 0;
	}
	for (int r = 0; r + 1 < G && sorted; ++r) {   // neighbours on either side of a shard boundary
		const size_t b = ranks[r + 1].first;
		if (b > 0 && b < n && host_kdf(hsrc + (b - 1) * kb, kb, dtype, order) > host_kdf(hsrc + b * kb, kb, dtype, order))
			sorted = false;
	}
	if (sorted) {                            // radix_sort.hpp:60-62
		*result = src;
		if (info)
			info->early_exit = 2;
		return RSX_OK;
	}
	const u64 key0 = host_kdf(hsrc, kb, dtype, order);   // radix_sort.hpp:64-70
	u32 cols[8], ncols = 0;
	for (u32 c = 0; c < kb; ++c)
		if (ghist[c * 256 + ((key0 >> (8 * c)) & 0xFF)] != n)
			cols[ncols++] = c;
	if (ncols == 0)
		return fail(RSX_EHIP, "rsx_sort_multi: unsorted input without a varying column");
	void *hres = (ncols & 1) ? aux : src;    // radix_sort.hpp:92
	if (info) {
		info->ncols = ncols;
		for (u32 i = 0; i < ncols; ++i)
			info->cols[i] = cols[i];
		info->result_in_aux = hres == aux;
	}

	// ---- destinations: contiguous digit ranges of the highest kept byte; matrix[s][d] keys go from shard s to rank d
	const u32 cs = cols[ncols - 1];
	uint8_t lut[256];
	choose_splitters_host(&ghist[cs * 256], G, lut);
	std::vector<u64> matrix((size_t)G * G, 0);
	for (int s = 0; s < G; ++s)
		for (int d = 0; d < 256; ++d)
			matrix[(size_t)s * G + lut[d]] += ranks[s].hist[cs * 256 + d];
	size_t running = 0;
	for (int d = 0; d < G; ++d) {
		ranks[d].out_first = running;
		ranks[d].n_recv = 0;
		for (int s = 0; s < G; ++s)
			ranks[d].n_recv += (size_t)matrix[(size_t)s * G + d];
		running += ranks[d].n_recv;
	}
	if (running != n)
		return fail(RSX_EHIP, "rsx_sort_multi: the count matrix sums to %zu, n = %zu", running, n);

	// ---- one stable pass by that byte per shard
	RSX_TRY(multi_phase(ranks, [&](MultiRank &k) -> int {
		if (k.count == 0)
			return RSX_OK;
		uint64_t top[256];
		RSX_TRY(rsx_msd_split_device(k.shard, k.part, k.count, dtype, order, (int)cs, top, k.stream));
		for (int d = 0; d < 256; ++d)
			if (top[d] != k.hist[cs * 256 + d])
				return fail(RSX_EHIP, "the split counted digit %d differently from the histogram", d);
		HIP_TRY(hipStreamSynchronize(k.stream));
		return RSX_OK;
	}));

	// ---- exchange (every rank pulls its digit range from every shard, in shard order), local sort, write-back
	char *hdst = (char *)hres;
	RSX_TRY(multi_phase(ranks, [&](MultiRank &k) -> int {
		if (k.n_recv == 0)
			return RSX_OK;
		const int d = (int)(&k - &ranks[0]);
		RSX_TRY(k.bufs->recv.ensure(k.n_recv * kb));   // sized from the count matrix, whatever the skew
		RSX_TRY(k.bufs->aux.ensure(k.n_recv * kb));
		k.recv = k.bufs->recv.p;
		k.aux = k.bufs->aux.p;
		for (int s = 0; s < G; ++s)
			if (matrix[(size_t)s * G + d])
				enable_peer(k.dev, ranks[s].dev);
		size_t off = 0;
		for (int s = 0; s < G; ++s) {
			const size_t cnt = (size_t)matrix[(size_t)s * G + d];
			if (cnt == 0)
				continue;
			size_t soff = 0;
			for (int e = 0; e < d; ++e)
				soff += (size_t)matrix[(size_t)s * G + e];
			const char *from = (const char *)ranks[s].part + soff * kb;
			if (ranks[s].dev == k.dev)
				HIP_TRY(hipMemcpyAsync((char *)k.recv + off * kb, from, cnt * kb, hipMemcpyDeviceToDevice, k.stream));
			else
				HIP_TRY(hipMemcpyPeerAsync((char *)k.recv + off * kb, k.dev, from, ranks[s].dev, cnt * kb, k.stream));
			off += cnt;
		}
		void *dres = nullptr;
		rsx_info li;
		RSX_TRY(rsx_sort_device(k.recv, k.aux, k.n_recv, dtype, order, k.stream, &dres, &li));
		HIP_TRY(hipMemcpyAsync(hdst + k.out_first * kb, dres, k.n_recv * kb, hipMemcpyDeviceToHost, k.stream));
		HIP_TRY(hipStreamSynchronize(k.stream));
		return RSX_OK;
	}));
	*result = hres;
	return RSX_OK;
}

