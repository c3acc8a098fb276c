// rsx_records.hpp: sorts of records -- opaque keys from the host (rsx_sort_rank_keys, rsx_sort_records) and a declared key
// field (rsx_sort_records_tagged[_device]); inside rsx.hip's extern "C" block -- part of librsx.so's host side; included by rsx.hip at the point where it used to stand (one translation unit:
// the kernels' instantiations are shared).  See rsx.hip for the context type, the error convention and the helpers used here.
#pragma once

int rsx_sort_rank_keys(const void *keys, size_t key_bytes, void *index_buffer, size_t n, size_t idx_bytes, void **result,
                       rsx_info *info)
{
	rsx_dtype dt;
	switch (key_bytes) {
	case 1: dt = RSX_U8; break;
	case 2: dt = RSX_U16; break;
	case 4: dt = RSX_U32; break;
	case 8: dt = RSX_U64; break;
	default: return fail(RSX_EINVAL, "rsx_sort_rank_keys: key_bytes must be 1, 2, 4 or 8");
	}
	return rsx_sort_rank(keys, index_buffer, n, dt, idx_bytes, RSX_ASCENDING, result, info);
}

int rsx_sort_records(void *src, void *aux, size_t n, size_t rec_bytes, const void *keys, size_t key_bytes, void **result,
                     rsx_info *info)
{
	rsx_dtype dt;
	switch (key_bytes) {
	case 1: dt = RSX_U8; break;
	case 2: dt = RSX_U16; break;
	case 4: dt = RSX_U32; break;
	case 8: dt = RSX_U64; break;
	default: return fail(RSX_EINVAL, "rsx_sort_records: key_bytes must be 1, 2, 4 or 8");
	}
	info_clear(info, dt);
	if (!result || !rec_bytes || (n && (!src || !aux || !keys)))
		return fail(RSX_EINVAL, "rsx_sort_records: bad argument");
	*result = src;
	if (n < 2) {
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(nullptr, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	const size_t wide = n > (1ull << 32) ? 8 : 4;
	RSX_TRY(c->keys[0].ensure(n * key_bytes));   // note: rank passes use keys[0]/keys[1] too; the uploaded keys live in recs[1]
	RSX_TRY(c->recs[1].ensure(n * key_bytes > n * rec_bytes ? n * key_bytes : n * rec_bytes));
	RSX_TRY(c->vals[0].ensure(2 * n * wide));
	HIP_TRY(hipMemcpyAsync(c->recs[1].p, keys, n * key_bytes, hipMemcpyHostToDevice, c->stream));
	void *dres = nullptr;
	rsx_info li;
	RSX_TRY(rsx_sort_rank_device(c->recs[1].p, c->vals[0].p, n, dt, wide, RSX_ASCENDING, nullptr, &dres, &li));
	if (info)
		*info = li;
	if (li.early_exit) {                     // pre-sorted by key: records stay where they are
		HIP_TRY(hipStreamSynchronize(c->stream));
		return RSX_OK;
	}
	// gather the records through the ranks (the keys in recs[1] are no longer needed)
	RSX_TRY(c->recs[0].ensure(n * rec_bytes));
	HIP_TRY(hipMemcpyAsync(c->recs[0].p, src, n * rec_bytes, hipMemcpyHostToDevice, c->stream));
	const uintptr_t al = (uintptr_t)rec_bytes;
	const unsigned grid = 2048, block = 256;
#define RSX_GATHER(WORD)                                                                                          \
	do {                                                                                                          \
		if (wide == 4)                                                                                            \
			hipLaunchKernelGGL((rsx_gather_kernel<WORD, u32>), dim3(grid), dim3(block), 0, c->stream,              \
			                   (WORD *)c->recs[1].p, (const WORD *)c->recs[0].p, (const u32 *)dres, (u64)n,         \
			                   (u32)(rec_bytes / sizeof(WORD)));                                                  \
		else                                                                                                      \
			hipLaunchKernelGGL((rsx_gather_kernel<WORD, u64>), dim3(grid), dim3(block), 0, c->stream,              \
			                   (WORD *)c->recs[1].p, (const WORD *)c->recs[0].p, (const u64 *)dres, (u64)n,         \
			                   (u32)(rec_bytes / sizeof(WORD)));                                                  \
	} while (0)
	if (al % 16 == 0)
		RSX_GATHER(uint4);
	else if (al % 8 == 0)
		RSX_GATHER(u64);
	else if (al % 4 == 0)
		RSX_GATHER(u32);
	else if (al % 2 == 0)
		RSX_GATHER(uint16_t);
	else
		RSX_GATHER(uint8_t);
#undef RSX_GATHER
	HIP_TRY(hipGetLastError());
	void *hres = li.result_in_aux ? aux : src;
	HIP_TRY(hipMemcpyAsync(hres, c->recs[1].p, n * rec_bytes, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	*result = hres;
	return RSX_OK;
}

// ---- records with a declared key: extraction, rank sort and gather on the device --------------------------------
namespace {

int gather_records(Ctx &c, void *d_out, const void *d_in, const void *d_idx, size_t idx_bytes, size_t n, size_t rec_bytes)
{
	const unsigned grid = 2048, block = 256;
	const uintptr_t al = (uintptr_t)rec_bytes | (uintptr_t)d_out | (uintptr_t)d_in;
#define RSX_GATHER(WORD)                                                                                              \
	do {                                                                                                              \
		if (idx_bytes == 4)                                                                                           \
			hipLaunchKernelGGL((rsx_gather_kernel<WORD, u32>), dim3(grid), dim3(block), 0, c.stream, (WORD *)d_out,      \
			                   (const WORD *)d_in, (const u32 *)d_idx, (u64)n, (u32)(rec_bytes / sizeof(WORD)));         \
		else                                                                                                          \
			hipLaunchKernelGGL((rsx_gather_kernel<WORD, u64>), dim3(grid), dim3(block), 0, c.stream, (WORD *)d_out,      \
			                   (const WORD *)d_in, (const u64 *)d_idx, (u64)n, (u32)(rec_bytes / sizeof(WORD)));         \
	} while (0)
	if (al % 16 == 0)
		RSX_GATHER(uint4);
	else if (al % 8 == 0)
		RSX_GATHER(u64);
	else if (al % 4 == 0)
		RSX_GATHER(u32);
	else if (al % 2 == 0)
		RSX_GATHER(uint16_t);
	else
		RSX_GATHER(uint8_t);
#undef RSX_GATHER
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// d_out = the records of d_in in stable key order (nothing is written when the keys are already sorted: li.early_exit)
int records_tagged_core(Ctx &c, const void *d_in, void *d_out, size_t n, size_t rec_bytes, size_t key_off, rsx_dtype dt,
                        rsx_order order, rsx_info *li)
{
	const size_t kb = dtype_size(dt);
	const size_t wide = n > (1ull << 32) ? 8 : 4;
	RSX_TRY(c.tkeys.ensure(n * kb));
	RSX_TRY(c.vals[0].ensure(2 * n * wide));
	RSX_DISPATCH_KT(dt, hipLaunchKernelGGL((rsx_extract_key_kernel<KT>), dim3(2048), dim3(256), 0, c.stream, (KT *)c.tkeys.p,
	                                      (const unsigned char *)d_in, (u64)n, (u32)rec_bytes, (u32)key_off));
	HIP_TRY(hipGetLastError());
	void *dres = nullptr;
	RSX_TRY(rsx_sort_rank_device(c.tkeys.p, c.vals[0].p, n, dt, wide, order, c.stream, &dres, li));
	if (li->early_exit)
		return RSX_OK;
	return gather_records(c, d_out, d_in, dres, wide, n, rec_bytes);
}

int tagged_args_ok(const char *who, size_t n, size_t rec_bytes, size_t key_off, rsx_dtype dt, const void *a, const void *b,
                   void **result)
{
	const size_t kb = dtype_size(dt);
	if (!kb)
		return fail(RSX_EINVAL, "%s: unknown key dtype", who);
	if (!result || !rec_bytes || key_off + kb > rec_bytes || rec_bytes > 0xFFFFFFFFull || (n && (!a || !b)))
		return fail(RSX_EINVAL, "%s: bad argument (the key must lie inside the record)", who);
	return RSX_OK;
}

}  // namespace

int rsx_sort_records_tagged_device(void *d_src, void *d_aux, size_t n, size_t rec_bytes, size_t key_offset, rsx_dtype key_dtype,
                                   rsx_order order, void *stream, void **result, rsx_info *info)
{
	info_clear(info, key_dtype);
	RSX_TRY(tagged_args_ok("rsx_sort_records_tagged_device", n, rec_bytes, key_offset, key_dtype, d_src, d_aux, result));
	*result = d_src;
	if (n < 2) {
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	rsx_info li;
	RSX_TRY(records_tagged_core(*c, d_src, d_aux, n, rec_bytes, key_offset, key_dtype, order, &li));
	if (info)
		*info = li;
	if (li.early_exit)
		return RSX_OK;
	if (li.result_in_aux)
		*result = d_aux;                      // radix_sort.hpp:92: odd number of kept columns
	else
		HIP_TRY(hipMemcpyAsync(d_src, d_aux, n * rec_bytes, hipMemcpyDeviceToDevice, c->stream));
	return RSX_OK;
}

int rsx_sort_records_tagged(void *src, void *aux, size_t n, size_t rec_bytes, size_t key_offset, rsx_dtype key_dtype,
                            rsx_order order, void **result, rsx_info *info)
{
	info_clear(info, key_dtype);
	RSX_TRY(tagged_args_ok("rsx_sort_records_tagged", n, rec_bytes, key_offset, key_dtype, src, aux, result));
	*result = src;
	if (n < 2) {
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(nullptr, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	RSX_TRY(c->recs[0].ensure(n * rec_bytes));
	RSX_TRY(c->recs[1].ensure(n * rec_bytes));
	HIP_TRY(hipMemcpyAsync(c->recs[0].p, src, n * rec_bytes, hipMemcpyHostToDevice, c->stream));
	rsx_info li;
	RSX_TRY(records_tagged_core(*c, c->recs[0].p, c->recs[1].p, n, rec_bytes, key_offset, key_dtype, order, &li));
	if (info)
		*info = li;
	if (!li.early_exit) {
		void *hres = li.result_in_aux ? aux : src;
		HIP_TRY(hipMemcpyAsync(hres, c->recs[1].p, n * rec_bytes, hipMemcpyDeviceToHost, c->stream));
		*result = hres;
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	return RSX_OK;
}

