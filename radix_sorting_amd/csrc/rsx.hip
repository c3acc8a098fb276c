// rsx.hip -- host side of librsx.so: the extern "C" surface declared in
// include/rsx.h on top of the kernels in rsx_kernels.hpp.
//
// The control flow mirrors rs_sort_main (radix_sort.hpp:31-93): histogram +
// pre-sorted test -> early exit -> column probe -> exclusive scan -> one stable
// scatter pass per kept column, ping-ponging two buffers -> "returned pointer".
// There is deliberately no CPU path here: if HIP cannot give us a gfx950
// device, every entry point fails with RSX_ENODEVICE.
#include "../../include/rsx.h"
#include "rsx_kernels.hpp"
#include "rsx_scatter2.hpp"
#include "rsx_small.hpp"
#include "rsx_hybrid.hpp"
#include "rsx_leaf16.hpp"
#include "rsx_pass16.hpp"
#include "rsx_pass32.hpp"
#include "rsx_leafc.hpp"
#include "rsx_logroute.hpp"
#include "rsx_pass64.hpp"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <functional>
#include <atomic>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

using namespace rsx;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			(void)hipGetLastError();                                                               \
			return fail(e_ == hipErrorOutOfMemory ? RSX_ENOMEM : RSX_EHIP, "%s failed: %s (%s:%d)", #expr, \
			            hipGetErrorString(e_), __FILE__, __LINE__);                                \
		}                                                                                          \
	} while (0)

#define RSX_TRY(expr)          \
	do {                       \
		int rc_ = (expr);      \
		if (rc_ != RSX_OK)     \
			return rc_;        \
	} while (0)

size_t dtype_size(int dtype)
{
	switch (dtype) {
	case RSX_U8: case RSX_I8: return 1;
	case RSX_U16: case RSX_I16: return 2;
	case RSX_U32: case RSX_I32: case RSX_F32: return 4;
	case RSX_U64: case RSX_I64: case RSX_F64: return 8;
	default: return 0;
	}
}

template <typename KT>
KdfArgs<KT> make_kdf(int dtype, int order)
{
	KdfArgs<KT> a;
	const KT high = (KT)((KT)1 << (sizeof(KT) * 8 - 1));
	const bool is_signed = dtype == RSX_I8 || dtype == RSX_I16 || dtype == RSX_I32 || dtype == RSX_I64;
	const bool is_float = dtype == RSX_F32 || dtype == RSX_F64;
	a.fmask = is_float ? (KT)~(KT)0 : (KT)0;
	a.sflip = (is_signed || is_float) ? high : (KT)0;
	a.desc = order == RSX_DESCENDING ? (KT)~(KT)0 : (KT)0;
	return a;
}

// ---- the RSX_* switches of the environment, read ONCE per process (rsx_reload_env() reads them again: tests) ----------------
// "set" switches are on when the variable exists, "=1" switches when its value starts with '1' (as documented in rsx.h).
std::atomic<u32> g_env_epoch{0};   // bumped by rsx_reload_env()
struct Env {
	bool host_register = false;      // RSX_HOST_REGISTER=1
	bool force_table_rank = false;   // RSX_FORCE_TABLE_RANK=1
	bool verify = false;             // RSX_VERIFY=1
	bool verify_whole = false;       // RSX_VERIFY=2: keys-only sorts check their whole result (sortedness + checksums), any route
	bool verify_inject = false;      // RSX_VERIFY_INJECT (set)
	bool no_hot = false;             // RSX_NO_HOT (set)
	bool elem_loads = false;         // RSX_ELEM_LOADS=1
	bool no_small_tiles = false;     // RSX_NO_SMALL_TILES (set)
	bool no_hybrid = false;          // RSX_NO_HYBRID=1
	bool no_small_sort = false;      // RSX_NO_SMALL_SORT (set)
	bool no_fill_runs = false;       // RSX_NO_FILL_RUNS (set)
	bool no_speculation = false;     // RSX_NO_SPECULATION (set)
	bool compact_bits = false;       // RSX_COMPACT_BITS=1
	bool no_narrow_keys = false;     // RSX_NO_NARROW_KEYS (set)
	bool no_host_small = false;      // RSX_NO_HOST_SMALL (set)
	bool no_fused_hist = false;      // RSX_NO_FUSED_HIST=1
	bool no_slack = false;           // RSX_NO_SLACK=1
	bool no_self_plan = false;       // RSX_NO_SELF_PLAN=1
	bool no_blind = false;           // RSX_NO_BLIND=1: every sort starts with the histogram
	unsigned blind_min_log2 = 0;     // RSX_BLIND_MIN_LOG2: keys-only sorts may skip the histogram from 2^this keys on (0: the measured floors)
	bool no_leaf_prefix = false;     // RSX_NO_LEAF_PREFIX=1: leaves of 8-byte keys sort by every column they have left (rsx_hybrid.hpp)
	bool no_leaf16w2k = false;       // RSX_NO_LEAF16W2K=1: slots of 1025 .. 2048 values take a 128-thread workgroup per leaf (rsx_leaf16_kernel) instead of a wave
	bool no_leaf16q = false;         // RSX_NO_LEAF16Q=1: slots of up to 256 values take a wave per leaf (rsx_leaf16w_kernel) instead of a row of sixteen lanes
	bool no_narrow_slots = false;    // RSX_NO_NARROW_SLOTS=1: the level-2 slots of 8-byte keys always hold whole keys (SegCtl::narrow)
	bool no_aux_slots = false;       // RSX_NO_AUX_SLOTS=1: the level-1 slots of a sort without a histogram all lie in scratch memory
	bool no_narrow1 = false;         // RSX_NO_NARROW_LEVEL1=1: the level-1 pass of 8-byte keys always writes whole keys (SegCtl::narrow stays below 2)
	bool no_dense_slots = false;     // RSX_NO_DENSE_SLOTS=1: the level-2 pass of a sort without a histogram writes whole keys
	bool force_dense_slots = false;  // RSX_DENSE_SLOTS=1: (kept for old scripts: two-byte slots are now written for every slot size rsx_leaf16_kernel takes)
	bool no_unstable = false;        // RSX_NO_UNSTABLE=1: the MSB passes of a sort without a histogram rank per wave (stable), as every other pass
	bool no_shift = false;           // RSX_NO_SHIFT=1: the MSB digits of a sort without a histogram are whole bytes (the two highest kept columns) always
	bool no_pass16 = false;          // RSX_NO_PASS16=1: the level-2 pass into two-byte slots is rsx_scatter2_kernel<..., KTO = u16, SEG> as in round 4 (rsx_pass16.hpp)
	unsigned pass16_wgs = 2;         // RSX_PASS16_WGS=1: ... one workgroup per CU (probe)
	bool no_packed_keys = false;     // RSX_NO_PACKED_KEYS=1: rank sorts without a histogram go by byte columns only (SegCtl::compact never set)
	bool no_pass32a = false;         // RSX_NO_PASS32A=1: the level-1 pass of such a sort is rsx_scatter2_kernel<..., SEG> with its look-back chain (rsx_pass32.hpp)
	unsigned pass32_min_mi = 0;      // RSX_PASS32_MIN_MI=k (probe): the level-1 atom pass from k Mi keys on (default: 52 Mi 4-byte keys, 24 Mi 8-byte keys)
	int pass32_prefetch = -1;        // RSX_PASS32_PREFETCH=0|1 (probe): rsx_pass32a_kernel requests a tile's keys while it writes the tile before (1) or when it starts on the tile (0, the default)
	bool no_pass16a = false;         // RSX_NO_PASS16A=1: ... whose runs are ragged (rsx_pass16_kernel) instead of whole 64-byte atoms (rsx_pass16a_kernel)
	unsigned pass16_dbg = 0;         // RSX_PASS16_DBG=1|2 (probe, WRONG OUTPUT): no stores / only whole aligned 64-byte atoms stored
	bool no_leafc = false;           // RSX_NO_LEAFC=1: no two-byte slots of more than 5120 values (rsx_leafc.hpp): sorts without a histogram of 4-byte keys end below 2^30 keys and their larger leaves sort whole keys, as in round 4
	unsigned force_leafc = 0;        // RSX_FORCE_LEAFC=1..6 (tests): two-byte slots of ANY size take the leaves of the large ones -- 1 the counting leaves at once, 2 / 3 / 4 / 5 / 6 rsx_leaf16_kernel's 10240- / 20480- / 6144- / 7680- / 15360-value shape and the counting leaves behind it
	bool no_leaf16 = false;          // RSX_NO_LEAF16=1: two-byte slots are sorted by rsx_leaf_sort_kernel (two LDS passes) as in round 3
	unsigned leaf16_maxbin = 25;     // RSX_LEAF16_MAXBIN (tests): leaves with a fuller bin go to rsx_leaf_sort_kernel (0: every leaf)
	unsigned leaf_grid = 65536;      // RSX_LEAF_GRID (probe): workgroups of a level-2 leaf launch (65536: one per table entry)
	unsigned two_level_min_log2 = 27; // RSX_TWO_LEVEL_MIN_LOG2: two MSB passes + leaves from 2^this keys on (tests: 22)
	bool no_odd_stride = false;      // RSX_NO_ODD_STRIDE=1: the level-1 slots of a sort without a histogram lie 1.25 means apart, rounded to 1 KiB, as in round 5
	unsigned cap1_pad_kib = 0;       // RSX_CAP1_PAD_KIB=k (probe): k KiB more per level-1 slot of a sort without a histogram
	unsigned probe = 0;              // RSX_PROBE=bits (measurements; results stay right): 1 the leaf table of a sort without a histogram in reverse slot order, 4 every device-scheduled sort as if hinted (rsx_sort_inplace_async_hint)
	bool no_pass64a = false;         // RSX_NO_PASS64A=1: the level-2 pass of 8-byte keys into four-byte slots is the chained rsx_scatter2_kernel of round 4 (rsx_pass64.hpp)
	bool no_log = false;             // RSX_NO_LOG=1: 8-byte keys never take the (bit length, mantissa) digits of rsx_logroute.hpp (rsx_info.hybrid never 6)
	bool log_leaf_big = false;       // RSX_LOG_LEAF_BIG=1 (tests): that route's leaves in the shape for 10240 values at every size
	bool pairs_leaf_big = false;     // RSX_PAIRS_LEAF_BIG=1 (tests): key + payload and rank sorts without a histogram: the leaves' shape for 10240 pairs at every size
	unsigned log_min_log2 = 0;       // RSX_LOG_MIN_LOG2: ... from 2^this keys on (tests: 20; default: from 24 Mi keys)
	void load()
	{
		auto is_set = [](const char *name) { return getenv(name) != nullptr; };
		auto is_one = [](const char *name) {
			const char *e = getenv(name);
			return e && e[0] == '1';
		};
		host_register = is_one("RSX_HOST_REGISTER");
		force_table_rank = is_one("RSX_FORCE_TABLE_RANK");
		verify = is_one("RSX_VERIFY");
		{
			const char *e = getenv("RSX_VERIFY");
			verify_whole = e && e[0] == '2';
		}
		verify_inject = is_set("RSX_VERIFY_INJECT");
		no_hot = is_set("RSX_NO_HOT");
		elem_loads = is_one("RSX_ELEM_LOADS");
		no_small_tiles = is_set("RSX_NO_SMALL_TILES");
		no_hybrid = is_one("RSX_NO_HYBRID");
		no_small_sort = is_set("RSX_NO_SMALL_SORT");
		no_fill_runs = is_set("RSX_NO_FILL_RUNS");
		no_speculation = is_set("RSX_NO_SPECULATION");
		compact_bits = is_one("RSX_COMPACT_BITS");
		no_narrow_keys = is_set("RSX_NO_NARROW_KEYS");
		no_host_small = is_set("RSX_NO_HOST_SMALL");
		no_fused_hist = is_one("RSX_NO_FUSED_HIST");
		no_slack = is_one("RSX_NO_SLACK");
		no_self_plan = is_one("RSX_NO_SELF_PLAN");
		no_blind = is_one("RSX_NO_BLIND");
		blind_min_log2 = 0;
		if (const char *e = getenv("RSX_BLIND_MIN_LOG2"))
			blind_min_log2 = (unsigned)std::max(22, std::min(30, atoi(e)));
		no_leaf_prefix = is_one("RSX_NO_LEAF_PREFIX");
		no_leaf16q = is_one("RSX_NO_LEAF16Q");
		no_narrow_slots = is_one("RSX_NO_NARROW_SLOTS");
		no_aux_slots = is_one("RSX_NO_AUX_SLOTS");
		no_narrow1 = is_one("RSX_NO_NARROW_LEVEL1");
		no_dense_slots = is_one("RSX_NO_DENSE_SLOTS");
		force_dense_slots = is_one("RSX_DENSE_SLOTS");
		no_leaf16 = is_one("RSX_NO_LEAF16");
		no_leafc = is_one("RSX_NO_LEAFC");
		no_leaf16w2k = is_one("RSX_NO_LEAF16W2K");
		if (const char *e = getenv("RSX_PASS32_MIN_MI"))
			pass32_min_mi = (unsigned)atoi(e);
		if (const char *e = getenv("RSX_PASS32_PREFETCH"))
			pass32_prefetch = e[0] == '1' ? 1 : 0;
		if (const char *e = getenv("RSX_FORCE_LEAFC"))
			force_leafc = (unsigned)atoi(e);
		no_pass16 = is_one("RSX_NO_PASS16");
		no_pass16a = is_one("RSX_NO_PASS16A");
		no_pass32a = is_one("RSX_NO_PASS32A");
		no_packed_keys = is_one("RSX_NO_PACKED_KEYS");
		pass16_wgs = 2;
		if (const char *e = getenv("RSX_PASS16_WGS"))
			pass16_wgs = atoi(e) == 1 ? 1u : 2u;
		pass16_dbg = 0;
		if (const char *e = getenv("RSX_PASS16_DBG"))
			pass16_dbg = (unsigned)atoi(e);
		no_shift = is_one("RSX_NO_SHIFT");
		no_unstable = is_one("RSX_NO_UNSTABLE");
		leaf16_maxbin = 25;
		if (const char *e = getenv("RSX_LEAF16_MAXBIN"))
			leaf16_maxbin = (unsigned)std::max(0, std::min(25, atoi(e)));
		leaf_grid = 65536;
		if (const char *e = getenv("RSX_LEAF_GRID"))
			leaf_grid = std::max(256, std::min(65536, atoi(e)));
		no_odd_stride = is_one("RSX_NO_ODD_STRIDE");
		cap1_pad_kib = 0;
		if (const char *e = getenv("RSX_CAP1_PAD_KIB"))
			cap1_pad_kib = (unsigned)std::max(0, std::min(65536, atoi(e)));
		probe = 0;
		if (const char *e = getenv("RSX_PROBE"))
			probe = (unsigned)atoi(e);
		no_pass64a = is_one("RSX_NO_PASS64A");
		no_log = is_one("RSX_NO_LOG");
		log_leaf_big = is_one("RSX_LOG_LEAF_BIG");
		pairs_leaf_big = is_one("RSX_PAIRS_LEAF_BIG");
		log_min_log2 = 0;
		if (const char *e = getenv("RSX_LOG_MIN_LOG2"))
			log_min_log2 = (unsigned)std::max(20, std::min(29, atoi(e)));
		two_level_min_log2 = 27;
		if (const char *e = getenv("RSX_TWO_LEVEL_MIN_LOG2")) {
			const int v = atoi(e);
			if (v >= 22 && v <= 30)
				two_level_min_log2 = (unsigned)v;
		}
	}
};
Env g_env;
std::once_flag g_env_once;
inline const Env &env()
{
	std::call_once(g_env_once, [] { g_env.load(); });
	return g_env;
}

// ---- the *_inplace_async entry points (enqueue only; the stream may be capturing a graph) mark their extent ----------------
// What the mark changes: a scratch buffer that would have to grow under a capture is an error instead of a hipFree / hipMalloc
// (DevBuf::ensure), and no slot array is ever released from inside such a call (blind_enqueue, pairs_blind_enqueue): a graph
// captured earlier on this context may still name it.
thread_local bool g_in_async = false;
thread_local hipStream_t g_async_stream = nullptr;
struct AsyncScope {
	bool prev;
	hipStream_t prev_stream;
	explicit AsyncScope(hipStream_t s) : prev(g_in_async), prev_stream(g_async_stream)
	{
		g_in_async = true;
		g_async_stream = s;
	}
	~AsyncScope()
	{
		g_in_async = prev;
		g_async_stream = prev_stream;
	}
	AsyncScope(const AsyncScope &) = delete;
	AsyncScope &operator=(const AsyncScope &) = delete;
};

// ---- a growable device allocation -------------------------------------------
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	bool external = false;   // a slice of a caller-owned workspace (rsx_sort_inplace_async_ws): never grown, never freed
	// Inside a *_inplace_async entry point (AsyncScope) the stream may be capturing: a buffer never grows under a capture -- the
	// graph would keep the old address, and hipFree / hipMalloc are not capturable.
	int ensure(size_t bytes)
	{
		if (bytes <= cap)
			return RSX_OK;
		if (external)
			return fail(RSX_EINVAL, "the caller's workspace is too small: %zu bytes needed where %zu were set aside "
			                        "(size it with rsx_workspace_bytes)", bytes, cap);
		if (g_in_async) {
			hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
			if (hipStreamIsCapturing(g_async_stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
				return fail(RSX_EINVAL, "a scratch buffer would have to grow (%zu -> %zu bytes) while the stream is capturing: run the "
				                        "sort once outside the capture, or use the *_ws entry points with a workspace of your own",
				            cap, bytes);
			(void)hipGetLastError();
		}
		// A buffer that GROWS is rounded up to an eighth of the power of two below its size (buffers of 2 MiB and more; smaller
		// ones get an eighth on top): a sort of slightly more keys than the last one -- the sub-ranges of a distributed sort, a
		// growing table -- finds room instead of paying hipFree + hipMalloc, and hipFree synchronises the device.  At most
		// 12.5 % above the request (round 4 gave GiB-sized slot arrays no headroom at all and re-allocated on every record
		// size).  A buffer's FIRST allocation is what was asked for (to 2 MiB): the four slot arrays of 2^28 pairs, a tile
		// above 1.25 GiB each, took 1.375 -- half a GiB for sorts that never come; sizes that do vary pay one re-allocation.
		size_t want = bytes + bytes / 8;
		if (!p && bytes >= ((size_t)2 << 20)) {
			want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		} else if (bytes >= ((size_t)2 << 20)) {
			size_t p2 = (size_t)1 << 21;
			while ((p2 << 1) <= bytes)
				p2 <<= 1;
			const size_t step = p2 / 8;
			want = (bytes + step - 1) / step * step;
		}
		// The new allocation is made BEFORE the old one goes: if it fails the old buffer (which a graph captured earlier may
		// still name) stays where it is; only then the old one is given up to make room.
		void *np = nullptr;
		hipError_t e = hipMalloc(&np, want);
		if (e != hipSuccess) {
			(void)hipGetLastError();
			want = bytes;
			e = hipMalloc(&np, want);
		}
		if (e != hipSuccess && p && !g_in_async) {
			(void)hipGetLastError();
			(void)hipFree(p);
			p = nullptr;
			cap = 0;
			e = hipMalloc(&np, want);
		}
		if (e != hipSuccess) {
			(void)hipGetLastError();
			return fail(RSX_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
		}
		if (p)
			(void)hipFree(p);
		p = np;
		cap = want;
		return RSX_OK;
	}
	void release()
	{
		if (p && !external)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	void borrow(void *ptr, size_t bytes)
	{
		p = ptr;
		cap = bytes;
		external = true;
	}
};

// ---- per (device, stream) context --------------------------------------------
struct Ctx {
	int device = -1;
	hipStream_t stream = nullptr;
	// fixed small state: [unsorted u32 | hotd 9 u32 | plan_done | pad 64][Plan 64][kept 16 u32 64]
	DevBuf small;
	DevBuf hist;        // counts, then exclusive offsets [key bytes][256] u64
	DevBuf hpart;       // the histogram kernel's per-workgroup rows [workgroups][key bytes][256] u32
	DevBuf status;      // [ticket u32, pad to 256 B][tiles * 256 status words]
	DevBuf keys[2];     // key ping-pong for rank sorts / host staging
	DevBuf vals[2];     // payload ping-pong for host staging / narrow-index rank
	DevBuf recs[2];     // record gather staging
	DevBuf tkeys;       // keys extracted from records (rsx_sort_records_tagged*)
	DevBuf ckeys;       // rank sorts: the keys' varying bits packed together (RSX_COMPACT_BITS)
	DevBuf joint;       // 2-byte keys: [65536 u32 counts][65537 u64 offsets] of the 16-bit digit (rsx_joint16_kernel)
	DevBuf seg;         // two-level sorts (rsx_hybrid.hpp): [SegCtl][per-bucket digit counts][status regions][leaf segments][tiles]
	size_t seg_hist_off = 0, seg_status_off = 0, seg_segtab_off = 0, seg_tiles_off = 0, seg_btile_off = 0, seg_redo_off = 0;
	SelfPlanArgs pass_sp{nullptr, nullptr, nullptr, nullptr, HybCaps{0, 0, 0, 0}};   // a self-planned pass 0 (SCATTER_SELF_PLAN)
	DevBuf gscan;       // [256] u64: the highest kept column's offsets from a self-planned pass 0 (for the leaves)
	// rsx_sort_inplace_async after an attempt without the histogram: the control block whose `mode` tells the histogram-first
	// kernels enqueued behind it that there is nothing left to do
	const SegCtl *pass_gate = nullptr;
	bool async_tried_blind = false;   // ... whether the last rsx_sort_inplace_async of this context enqueued such an attempt
	bool ws_blind = false;            // a context in a caller's workspace that has room for the slots of a sort without a histogram (borrow_ctx)
	bool boff_forget = false;   // rsx_reload_env since the last attempt: SegCtl::boff_* are zeroed before the next one
	u32 hints = 0;                    // what the caller of the sort being enqueued has said about its keys (rsx_sort_inplace_async_hint)
	bool async_small = false;         // ... or was the one-launch sort of a small array (rsx_async_route: 0, whatever the device's words say)
	const void *pass_alt = nullptr;   // rsx_sort_rank_inplace_async: the second work copy of the keys (SCATTER_RANK_ASYNC passes)
	DevBuf vsum;        // RSX_VERIFY=2: [descents, sum, mix] of the input and of the result
	DevBuf vasync;      // RSX_VERIFY: mismatches found in device-scheduled passes, kept until rsx_verify_poll / the next blocking sort
	DevBuf slack_v;     // ... the payloads' slots (key + payload and rank sorts)
	DevBuf slack;       // two-level sorts, slack attempt: 65536 slots of slack_cap keys (+ a tile of padding)
	u32 slack_cap = 0;
	u32 slack_mean = 0;   // the mean number of keys of a level-2 slot of the sort being enqueued (n / 65536: pass16a_wanted)
	DevBuf slack1_v;    // ... and of as many payloads (pairs_blind)
	DevBuf slack1;      // sorts without a histogram (sort_keys_blind): the level-1 pass's 256 slots of slack1_cap keys
	DevBuf logb;        // rsx_logroute.hpp: [LogCtl][LogTabs][level-2 cursors 2 x 65536][level-2 tiles]
	DevBuf logslots;    // ... the level-2 slots (four bytes per key)
	LogCtl *host_logctl = nullptr;   // pinned: the control block as the device left it
	hipEvent_t log_ev = nullptr;
	u32 slack1_cap = 0;
	u32 slack1_lo = 0;  // ... of which the first slack1_lo lie in the caller's second buffer (keys-only sorts; 0: all in slack1)
	bool narrow1 = false;   // 8-byte keys: the forms that keep low words in the level-1 slots are enqueued too (SegCtl::narrow == 2 picks them)
	// ... sorts to go before the next attempt, doubled by every attempt that is called off; per kind of sort (4- / 8-byte keys,
	// rank sorts, keys + payload): what one kind's inputs look like says nothing about another's
	u32 blind_skip[4] = {0, 0, 0, 0}, blind_backoff[4] = {0, 0, 0, 0};
	u32 log_skip = 0, log_backoff = 0;       // ... and the same for the attempts by (bit length, mantissa) digits (sort_keys_log): a refused or lost one costs 65 us and more
	bool blind_no_room = false;              // the slots could not be allocated once: not asked for again (until rsx_reload_env)
	u32 env_epoch = 0;                       // ... forgotten when rsx_reload_env() has run since
	SegCtl *host_segctl = nullptr, *dev_host_segctl = nullptr;   // pinned, written by rsx_seg_plan_kernel
	hipEvent_t seg_ev = nullptr;
	Plan *host_plan = nullptr;   // pinned, written by the kernels themselves (dev_host_plan: its device address)
	Plan *dev_host_plan = nullptr;
	hipEvent_t plan_ev = nullptr;   // recorded behind the plan's copy to the host
	u64 *host_hist = nullptr;    // pinned, 256 u64
	// Small sorts of host buffers (rsx_sort, rsx_sort_rank on arrays the one-launch kernels take): pinned, mapped staging the
	// kernel reads and writes over PCIe itself -- one launch and one synchronisation instead of two copies around them.
	char *hstage = nullptr, *hstage_dev = nullptr;
	static constexpr size_t HSTAGE_BYTES = 5 * (size_t)SMALL_SORT_BYTES;   // keys, keys, and two halves of 8-byte indices

	bool fast = false;           // rsx_scatter2_kernel allowed on this device (LDS atomic order verified)
	// The reference is re-entrant (concurrent calls on disjoint buffers are safe); here calls that share a context
	// (same device and stream) share its workspace, so every entry point holds this for its duration.
	std::recursive_mutex mu;

	// The library's own contexts hold TWO sets of flags and histograms and alternate between them (`gen`): small sorts zero
	// the set of the next sort inside this sort's histogram kernel instead of launching a kernel for it (plan_phase).  A
	// context in a caller's workspace (external) has one set.
	u32 gen = 0;
	static constexpr size_t SMALL_BYTES = 256, HIST_SET_BYTES = 8 * 256 * sizeof(u64);
	char *small_set() const { return (char *)small.p + (small.external ? 0 : gen * SMALL_BYTES); }
	char *small_set_other() const { return (char *)small.p + (gen ^ 1u) * SMALL_BYTES; }
	u64 *ghist() const { return (u64 *)((char *)hist.p + (hist.external ? 0 : gen * HIST_SET_BYTES)); }
	u64 *ghist_other() const { return (u64 *)((char *)hist.p + (gen ^ 1u) * HIST_SET_BYTES); }
	u32 *unsorted() const { return (u32 *)small_set(); }
	u32 *plan_done() const { return (u32 *)(small_set() + 52); }   // blocks of rsx_plan_kernel that are through
	u64 *verify_bad() const { return (u64 *)(small_set() + 56); }  // RSX_VERIFY: mismatches found by rsx_verify_tile_kernel
	u32 *hotd() const { return (u32 *)(small_set() + 16); }   // [8] hot digits per column + [1] valid bits (rsx_plan_kernel)
	Plan *plan() const { return (Plan *)(small_set() + 64); }
	u32 *kept() const { return (u32 *)(small_set() + 128); }
	u32 *colmax() const { return (u32 *)(small_set() + 192); }   // [8] largest bin per column (rsx_plan_kernel, kept[16..])

	int init()
	{
		RSX_TRY(small.ensure(2 * SMALL_BYTES));
		RSX_TRY(hist.ensure(2 * HIST_SET_BYTES));
		RSX_TRY(gscan.ensure(256 * sizeof(u64)));
		HIP_TRY(hipMemset(small.p, 0, 2 * SMALL_BYTES));   // (both sets start out zeroed: see `gen`)
		HIP_TRY(hipMemset(hist.p, 0, 2 * HIST_SET_BYTES));
		if (!host_plan)
		{
			HIP_TRY(hipHostMalloc((void **)&host_plan, sizeof(Plan), hipHostMallocMapped));
			HIP_TRY(hipHostGetDevicePointer((void **)&dev_host_plan, host_plan, 0));
		}
		if (!host_hist)
			HIP_TRY(hipHostMalloc((void **)&host_hist, 256 * sizeof(u64), hipHostMallocDefault));
		if (!host_segctl) {
			HIP_TRY(hipHostMalloc((void **)&host_segctl, sizeof(SegCtl), hipHostMallocMapped));
			HIP_TRY(hipHostGetDevicePointer((void **)&dev_host_segctl, host_segctl, 0));
		}
		return RSX_OK;
	}
	int ensure_hstage()
	{
		if (!hstage) {
			HIP_TRY(hipHostMalloc((void **)&hstage, HSTAGE_BYTES, hipHostMallocMapped));
			HIP_TRY(hipHostGetDevicePointer((void **)&hstage_dev, hstage, 0));
		}
		return RSX_OK;
	}
	void release()
	{
		if (hstage)
			(void)hipHostFree(hstage);
		hstage = hstage_dev = nullptr;
		small.release();
		hist.release();
		hpart.release();
		status.release();
		for (int i = 0; i < 2; ++i) {
			keys[i].release();
			vals[i].release();
			recs[i].release();
		}
		tkeys.release();
		ckeys.release();
		joint.release();
		seg.release();
		slack.release();
		slack1.release();
		logb.release();
		logslots.release();
		if (host_logctl)
			(void)hipHostFree(host_logctl);
		host_logctl = nullptr;
		if (log_ev)
			(void)hipEventDestroy(log_ev);
		log_ev = nullptr;
		slack1_v.release();
		slack_v.release();
		vasync.release();
		vsum.release();
		gscan.release();
		if (host_segctl)
			(void)hipHostFree(host_segctl);
		host_segctl = dev_host_segctl = nullptr;
		if (seg_ev)
			(void)hipEventDestroy(seg_ev);
		seg_ev = nullptr;
		if (plan_ev)
			(void)hipEventDestroy(plan_ev);
		plan_ev = nullptr;
		if (host_plan)
			(void)hipHostFree(host_plan);
		if (host_hist)
			(void)hipHostFree(host_hist);
		host_plan = nullptr;
		host_hist = nullptr;
	}
};

std::mutex g_mu;
std::map<std::pair<int, void *>, Ctx *> g_ctx;
std::map<int, int> g_lds_order_ok;   // device -> result of lds_order_selfcheck (1 ok, 0 not)

// ---- RSX_HOST_REGISTER=1 (a measurement switch): the caller's host buffers are page-locked (hipHostRegister) FOR THE
// DURATION OF THE CALL, so that the copies go by DMA straight from / to them instead of through the runtime's bounce buffers.
// (Round 2 kept registrations cached per pointer across calls: a buffer freed and reallocated at the same address then
// reused a stale mapping, and evicting an entry could pull the registration from under another thread's copy.  Measured
// gain of keeping buffers registered: 1-2 %; a caller that wants it registers its own buffers with
// hipHostRegister -- the library copies from registered memory as it finds it.)
struct HostRegScope {
	void *p = nullptr;
	HostRegScope(void *ptr, size_t bytes)
	{
		if (!env().host_register || bytes < ((size_t)1 << 20))
			return;
		if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess)
			p = ptr;
		(void)hipGetLastError();   // (a buffer that cannot be registered -- or already is -- is copied as it is)
	}
	~HostRegScope()
	{
		if (p)
			(void)hipHostUnregister(p);
	}
	HostRegScope(const HostRegScope &) = delete;
	HostRegScope &operator=(const HostRegScope &) = delete;
};

// ---- optional HIP-event bracketing of the kernels (rsx_profile_begin/end) ------
struct ProfRec {
	int kind;   // 0 histogram, 1 scatter, 2 leaves (rsx_hybrid.hpp), 3 passes that write narrowed keys into slots
	hipEvent_t start, stop;
	u64 bytes;
	hipStream_t stream;
	bool called_off;   // launches of an attempt the device called off (or of a route the plan did not choose): they returned at
	                   // once or their output was discarded -- their time is booked apart, their bytes are not booked at all
	// device-scheduled sorts (rsx_sort_inplace_async): nobody reads a verdict back while the sort is enqueued, so the record names
	// a pinned word that receives SegCtl::mode behind the attempt (prof_verdict_slot) and which value makes it count:
	// valid_if 1 -- the attempt's own launches: SEG_MODE_LEAVES; 2 -- the gated histogram-first launches behind it: anything else
	const u32 *verdict = nullptr;
	int valid_if = 0;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;

// What a sort books is what the DEVICE chose.  Kernels are enqueued before the host knows the route (a sort without a histogram
// may be called off by its sample; leaves are launched in every shape the plan may ask for); once the host has the verdict it
// takes the records of the launches that did nothing out of the byte count (prof_called_off) or corrects their bytes
// (prof_rebook: 8-byte keys whose level-2 slots the sample narrowed to four bytes).  prof_mark() = where this call's records start.
size_t prof_mark()
{
	if (!g_prof_on)
		return 0;
	std::lock_guard<std::mutex> lock(g_prof_mu);
	return g_prof.size();
}
void prof_called_off(size_t mark, hipStream_t s, int kind = -1)
{
	if (!g_prof_on)
		return;
	std::lock_guard<std::mutex> lock(g_prof_mu);
	for (size_t i = mark; i < g_prof.size(); ++i)
		if (g_prof[i].stream == s && (kind < 0 || g_prof[i].kind == kind))
			g_prof[i].called_off = true;
}
// (the LAST record of `kind` since the mark: a sort's level-1 and level-2 passes are both kind 1, in that order)
void prof_rebook(size_t mark, hipStream_t s, int kind, u64 bytes, int new_kind = -1)
{
	if (!g_prof_on)
		return;
	std::lock_guard<std::mutex> lock(g_prof_mu);
	for (size_t i = g_prof.size(); i > mark; --i)
		if (g_prof[i - 1].stream == s && g_prof[i - 1].kind == kind && !g_prof[i - 1].called_off) {
			g_prof[i - 1].bytes = bytes;
			if (new_kind >= 0)
				g_prof[i - 1].kind = new_kind;
			break;
		}
}

// a pinned word for one device-scheduled sort's verdict (4096 per profile window; none left: the records stay as they are)
u32 *g_prof_vblock = nullptr;
size_t g_prof_vnext = 0;
u32 *prof_verdict_slot()
{
	std::lock_guard<std::mutex> lock(g_prof_mu);
	if (!g_prof_vblock && hipHostMalloc((void **)&g_prof_vblock, 4096 * sizeof(u32), hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		g_prof_vblock = nullptr;
		return nullptr;
	}
	if (g_prof_vnext >= 4096)
		return nullptr;
	u32 *p = g_prof_vblock + g_prof_vnext++;
	*p = 0;
	return p;
}
void prof_tag(size_t from, size_t to, hipStream_t s, const u32 *verdict, int valid_if)
{
	std::lock_guard<std::mutex> lock(g_prof_mu);
	for (size_t i = from; i < to && i < g_prof.size(); ++i)
		if (g_prof[i].stream == s) {
			g_prof[i].verdict = verdict;
			g_prof[i].valid_if = valid_if;
		}
}
// ... around the attempt and the gated launches of a device-scheduled sort (`attempted`: an attempt was enqueued at all)
struct ProfAsyncVerdict {
	size_t m0 = 0, m1 = 0;
	u32 *slot = nullptr;
	hipStream_t stream;
	explicit ProfAsyncVerdict(hipStream_t s) : stream(s) { m0 = prof_mark(); }
	void attempt_enqueued(const SegCtl *ctl)
	{
		if (!g_prof_on)
			return;
		slot = prof_verdict_slot();
		if (slot && hipMemcpyAsync(slot, &ctl->mode, sizeof(u32), hipMemcpyDeviceToHost, stream) != hipSuccess) {
			(void)hipGetLastError();
			slot = nullptr;
		}
		m1 = prof_mark();
	}
	void gated_enqueued()
	{
		if (!g_prof_on || !slot)
			return;
		const size_t m2 = prof_mark();
		prof_tag(m0, m1, stream, slot, 1);
		prof_tag(m1, m2, stream, slot, 2);
	}
};

struct ProfScope {
	bool on;
	ProfRec rec;
	hipStream_t stream;
	ProfScope(int kind, u64 bytes, hipStream_t s) : on(g_prof_on), stream(s)
	{
		if (!on)
			return;
		rec.kind = kind;
		rec.bytes = bytes;
		rec.stream = s;
		rec.called_off = false;
		if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) {
			on = false;
			return;
		}
		(void)hipEventRecord(rec.start, stream);
	}
	~ProfScope()
	{
		if (!on)
			return;
		(void)hipEventRecord(rec.stop, stream);
		std::lock_guard<std::mutex> lock(g_prof_mu);
		g_prof.push_back(rec);
	}
};
int g_devcount = -2;   // -2: not probed

int probe_devices()
{
	if (g_devcount != -2)
		return g_devcount;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		n = 0;
	}
	int usable = 0;
	for (int d = 0; d < n; ++d) {
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
			++usable;
		else
			(void)hipGetLastError();
	}
	g_devcount = usable;
	return usable;
}

// rsx_scatter2_kernel ranks keys with returning LDS atomics and needs them to resolve same-address lanes
// in lane order.  gfx950 does, but that is an observed property, not a documented one: verify it once
// per device (about a millisecond) and otherwise stay on the table-based ranking of rsx_scatter_kernel.
int lds_order_selfcheck(int dev)
{
	auto it = g_lds_order_ok.find(dev);
	if (it != g_lds_order_ok.end())
		return it->second;
	int ok = 0;
	u64 *d_bad = nullptr;
	if (!env().force_table_rank && hipMalloc((void **)&d_bad, sizeof(u64)) == hipSuccess) {
		u64 bad = ~0ull;
		if (hipMemset(d_bad, 0, sizeof(u64)) == hipSuccess) {
			// two shapes: eight waves of bare atomics on collision-heavy digits, and the production shape of
			// rsx_scatter2_kernel (16 waves, eight atomics in flight, staging stores and 16-byte rows between them)
			hipLaunchKernelGGL(rsx_lds_order_check_kernel, dim3(1024), dim3(512), 0, 0, d_bad, 0x9E3779B9u, 512);
			hipLaunchKernelGGL(rsx_lds_order_check2_kernel, dim3(512), dim3(1024), 0, 0, d_bad, 0x85EBCA6Bu, 256);
			if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
			    hipMemcpy(&bad, d_bad, sizeof(u64), hipMemcpyDeviceToHost) == hipSuccess)
				ok = bad == 0;
		}
		(void)hipFree(d_bad);
	}
	(void)hipGetLastError();
	g_lds_order_ok[dev] = ok;
	return ok;
}

// ---- rsx_capture_histogram: the caller-supplied Hist of rs_sort_main (radix_sort.hpp:28-33) -------------------------
// Armed per thread; the next sort of that thread that runs the histogram kernels writes the raw per-column digit counts
// there (hist[256 j + d]) and disarms.  The counts are recovered from the exclusive offsets the plan kernel leaves in the
// workspace (every column is scanned, kept or not): count[d] = offset[d + 1] - offset[d], the last one n - offset[255].
thread_local u64 *g_capture_dst = nullptr;
thread_local size_t g_capture_entries = 0;

inline bool capture_armed() { return g_capture_dst != nullptr; }

int capture_hist(Ctx &c, size_t n, size_t kb)
{
	if (!g_capture_dst)
		return RSX_OK;
	u64 *dst = g_capture_dst;
	const size_t entries = g_capture_entries;
	g_capture_dst = nullptr;
	g_capture_entries = 0;
	if (entries < 256 * kb)
		return fail(RSX_EINVAL, "rsx_capture_histogram: room for %zu entries, the sort has %zu", entries, 256 * kb);
	std::vector<u64> off(256 * kb);
	HIP_TRY(hipMemcpyAsync(off.data(), c.ghist(), 256 * kb * sizeof(u64), hipMemcpyDeviceToHost, c.stream));
	HIP_TRY(hipStreamSynchronize(c.stream));
	for (size_t j = 0; j < kb; ++j)
		for (size_t d = 0; d < 256; ++d)
			dst[256 * j + d] = (d == 255 ? (u64)n : off[256 * j + d + 1]) - off[256 * j + d];
	return RSX_OK;
}

// RSX_VERIFY=1 (read once per process): after every host-scheduled scatter pass of the fast kernel one pseudo-randomly
// chosen tile is re-ranked without LDS atomics (rsx_verify_tile_kernel) and compared with what the pass wrote; a
// mismatch fails the call with RSX_EVERIFY.  Passes are then serialised by the check's read-back and no pass is
// speculative; the *_inplace_async entry points, which never synchronise, are not verified.
bool verify_mode() { return env().verify; }
u32 g_verify_seq = 0;

int get_ctx(void *stream, Ctx **out)
{
	std::lock_guard<std::mutex> lock(g_mu);
	if (probe_devices() <= 0)
		return fail(RSX_ENODEVICE, "no gfx950 (MI355X) device visible to HIP; this library has no CPU path");
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	auto key = std::make_pair(dev, stream);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) {
		Ctx *c = new Ctx();
		c->device = dev;
		c->stream = (hipStream_t)stream;
		c->fast = lds_order_selfcheck(dev) != 0;
		int rc = c->init();
		if (rc != RSX_OK) {
			c->release();
			delete c;
			return rc;
		}
		it = g_ctx.emplace(key, c).first;
	}
	*out = it->second;
	return RSX_OK;
}

void info_clear(rsx_info *info, int dtype)
{
	if (!info)
		return;
	memset(info, 0, sizeof(*info));
	info->key_bytes = (uint32_t)dtype_size(dtype);
}

void info_from_plan(rsx_info *info, const Plan &p)
{
	if (!info)
		return;
	info->ncols = p.ncols;
	for (u32 i = 0; i < p.ncols && i < 8; ++i)
		info->cols[i] = p.cols[i];
}

// tiles per super-tile: as many as keeps at least ~2048 super-tiles in flight, at most 8
u32 choose_tps(size_t n, size_t tile)
{
	const u64 tiles = (n + tile - 1) / tile;
	u64 tps = tiles / 2048;
	if (tps > 8)
		tps = 8;
	if (tps < 1)
		tps = 1;
	return (u32)tps;
}

// ---- phase 1: histogram + plan (radix_sort.hpp:48-80) --------------------------
template <typename KT>
int launch_hist(Ctx &c, const KT *d_src, size_t n, KdfArgs<KT> ka, u64 *d_hist, u32 *d_unsorted, u32 colmask = ~0u,
                const HistFuse *fuse = nullptr)
{
	typedef HistCfg<KT> C;
	const u64 per_block = (u64)C::BLOCK * C::U * C::VEC;     // elements one workgroup covers per sweep
	u64 blocks = (n + per_block - 1) / per_block;
	if (blocks > 512)                                         // 256 CUs x 2 workgroups of 1024 threads
		blocks = 512;
	if (blocks < 1)
		blocks = 1;
	const u32 cols256 = (u32)sizeof(KT) * 256;
	RSX_TRY(c.hpart.ensure((size_t)blocks * cols256 * sizeof(u32)));
	ProfScope prof(0, (u64)n * sizeof(KT), c.stream);
	// up to 128 workgroups add their counts to the histogram themselves (one launch and its gap less: 7 of the 62 us of
	// a 10^5-key sort); beyond that the rows are summed by a kernel of their own
	const bool direct = blocks <= 128;
	const u32 *gate = c.pass_gate ? &c.pass_gate->mode : nullptr;
	HistFuse nofuse{};
	nofuse.gate = gate;
	if (fuse) {
		// the kernel also zeroes what the caller names (FUSED, rsx_hist.hpp)
		if (ka.fmask == 0 && ka.sflip == 0 && ka.desc == 0)
			hipLaunchKernelGGL((rsx_hist_kernel<KT, C, HIST_PLAIN, true>), dim3((unsigned)blocks), dim3(C::BLOCK), 0, c.stream, d_src,
			                   (u64)n, (u32 *)c.hpart.p, d_unsorted, ka, colmask, direct ? d_hist : (u64 *)nullptr, *fuse);
		else
			hipLaunchKernelGGL((rsx_hist_kernel<KT, C, HIST_GENERIC, true>), dim3((unsigned)blocks), dim3(C::BLOCK), 0, c.stream, d_src,
			                   (u64)n, (u32 *)c.hpart.p, d_unsorted, ka, colmask, direct ? d_hist : (u64 *)nullptr, *fuse);
	} else
	// keys that are their own KDF (unsigned, ascending) take the instantiation without the KDF arithmetic
	if (ka.fmask == 0 && ka.sflip == 0 && ka.desc == 0)
		hipLaunchKernelGGL((rsx_hist_kernel<KT, C, HIST_PLAIN>), dim3((unsigned)blocks), dim3(C::BLOCK), 0, c.stream, d_src, (u64)n,
		                   (u32 *)c.hpart.p, d_unsorted, ka, colmask, direct ? d_hist : (u64 *)nullptr, nofuse);
	else
		hipLaunchKernelGGL((rsx_hist_kernel<KT, C, HIST_GENERIC>), dim3((unsigned)blocks), dim3(C::BLOCK), 0, c.stream, d_src, (u64)n,
		                   (u32 *)c.hpart.p, d_unsorted, ka, colmask, direct ? d_hist : (u64 *)nullptr, nofuse);
	if (!direct)
		hipLaunchKernelGGL(rsx_hist_reduce_kernel, dim3((unsigned)sizeof(KT), HIST_REDUCE_SPLIT), dim3(256), 0, c.stream,
		                   (const u32 *)c.hpart.p, d_hist, (u32)blocks, cols256, gate);
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

template <typename KT>
int plan_phase(Ctx &c, const KT *d_src, size_t n, KdfArgs<KT> ka, Plan *out, size_t status_total = 0,
               HybCaps caps = HybCaps{0, 0, 0, 0}, bool fuse_ok = false, bool *self_plan = nullptr)
{
	// self_plan (in: the caller would like pass 0 to derive the plan itself, SCATTER_SELF_PLAN; out: whether it has to):
	// then only the histogram is enqueued here -- no plan kernel, no event
	const size_t hist_bytes = sizeof(KT) * 256 * sizeof(u64);
	if (c.hist.external)
		RSX_TRY(c.hist.ensure(hist_bytes));
	// (fuse_ok: the blocking entry points only.  A captured graph replays the SAME launch, so it cannot alternate between
	// the two sets of flags: the device-scheduled *_async sorts keep the separate zeroing launch.)
	if (fuse_ok && status_total && !out && !c.small.external && !env().no_fused_hist) {
		// Small arrays: the histogram kernel also zeroes (no launch for that).  The set of flags / histogram this sort uses
		// was zeroed by the previous sort (or at start-up); this sort's histogram kernel zeroes the other set for the next
		// one, and the status words of its own passes (which run after it).  One workgroup then makes the plan.
		RSX_TRY(c.status.ensure(status_total));
		c.gen ^= 1u;
		HistFuse f{};
		f.z0 = (u32x4 *)c.small_set_other();
		f.n0 = Ctx::SMALL_BYTES / 16;
		f.z1 = (u32x4 *)c.ghist_other();
		f.n1 = Ctx::HIST_SET_BYTES / 16;
		f.z2 = (u32x4 *)c.status.p;
		f.n2 = status_total / 16;
		RSX_TRY(launch_hist<KT>(c, d_src, n, ka, c.ghist(), c.unsorted(), ~0u, &f));
		if (self_plan && *self_plan)
			return RSX_OK;
		hipLaunchKernelGGL((rsx_plan_all_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, d_src, (u64)n, c.ghist(), ka, c.kept(),
		                   c.hotd(), (const u32 *)c.unsorted(), c.plan(), c.dev_host_plan, caps);
		HIP_TRY(hipGetLastError());
		if (!c.plan_ev)
			HIP_TRY(hipEventCreateWithFlags(&c.plan_ev, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(c.plan_ev, c.stream));
		return RSX_OK;
	}
	if (self_plan)
		*self_plan = false;
	if (status_total) {
		// flags, histogram and the status words of every pass of this sort in one launch
		RSX_TRY(c.status.ensure(status_total));
		const u64 total16 = (256 + hist_bytes + status_total) / 16;
		const unsigned blocks = (unsigned)std::min<u64>((total16 + 255) / 256, 2048);
		hipLaunchKernelGGL(rsx_zero3_kernel, dim3(blocks), dim3(256), 0, c.stream, (u32x4 *)c.small_set(), (u64)(256 / 16),
		                   (u32x4 *)c.ghist(), (u64)(hist_bytes / 16), (u32x4 *)c.status.p, (u64)(status_total / 16),
		                   c.pass_gate ? &c.pass_gate->mode : (const u32 *)nullptr);
	} else {
		HIP_TRY(hipMemsetAsync(c.ghist(), 0, hist_bytes, c.stream));
		HIP_TRY(hipMemsetAsync(c.small_set(), 0, 256, c.stream));
	}
	RSX_TRY(launch_hist<KT>(c, d_src, n, ka, c.ghist(), c.unsorted()));
	hipLaunchKernelGGL((rsx_plan_all_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, d_src, (u64)n, c.ghist(), ka, c.kept(),
	                   c.hotd(), (const u32 *)c.unsorted(), c.plan(), c.dev_host_plan, caps,
	                   c.pass_gate ? &c.pass_gate->mode : (const u32 *)nullptr);
	HIP_TRY(hipGetLastError());
	if (!out) {   // the caller enqueues more work and collects the plan with plan_wait()
		if (c.small.external)
			return RSX_OK;   // (a caller-owned workspace: nobody waits for this plan on the host)
		if (!c.plan_ev)
			HIP_TRY(hipEventCreateWithFlags(&c.plan_ev, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(c.plan_ev, c.stream));
		return RSX_OK;
	}
	HIP_TRY(hipStreamSynchronize(c.stream));
	*out = *c.host_plan;
	return RSX_OK;
}

int plan_wait(Ctx &c, Plan *out)
{
	HIP_TRY(hipEventSynchronize(c.plan_ev));
	*out = *c.host_plan;
	return RSX_OK;
}

// the flags of a pass over a column with hot digits (Plan::hot): the HOT kernel + the column, for its hotd word
inline u32 hot_flags(u32 hotmask, u32 col)
{
	if (env().no_hot)   // (diagnostic: the plain kernels on columns with hot digits)
		return 0u;
	return (hotmask >> col & 1u) ? ((u32)SCATTER_HOT | (col << SCATTER_COL_SHIFT)) : 0u;
}

// ---- phase 2: one scatter pass (radix_sort.hpp:83-90) -----------------------------
// gbase[digit]: exclusive offset of the digit for this pass's column
template <typename KT, typename VT, typename C2, typename KTO = KT>
int launch_scatter2(Ctx &c, const KT *kin, KTO *kout, const VT *vin, VT *vout, size_t n, u32 shift, const u64 *gbase,
                    KdfArgs<KT> ka, u32 flags, const Plan *dplan, int region, u32 pass_index, u32 oshift = 0)
{
	const u64 tiles = (n + C2::TILE - 1) / C2::TILE;
	const u32 tps = (u32)C2::TPS;   // 1: a tile is its own super-tile (32-bit cells leave no LDS for a second tile's counts)
	const bool wide = n >= (1ull << 30);   // counter width by n, as radix_sort.hpp:102-114 does
	const size_t st_bytes = 256 + tiles * 256 * (wide ? 8 : 4);
	char *base = (char *)c.status.p;
	if (region < 0) {   // own region, zeroed here
		RSX_TRY(c.status.ensure(st_bytes));
		HIP_TRY(hipMemsetAsync(c.status.p, 0, st_bytes, c.stream));
		base = (char *)c.status.p;
	} else {            // region `region` of a buffer the caller has sized (status_bytes) and zeroed (plan_phase)
		base += (size_t)region * st_bytes;
	}
	u32 *ticket = (u32 *)base;
	void *st = base + 256;
	ProfScope prof(1, (u64)n * (sizeof(KT) + sizeof(KTO) + 2 * val_bytes<VT>::value), c.stream);
	const dim3 grid((unsigned)tiles);
	// keys that are their own KDF (unsigned ascending, no bucket table) take the kernel without the KDF arithmetic;
	// columns with a hot digit (Plan::hot) take the instantiation that tests every round for a wave-uniform digit
	const bool integer = val_bytes<VT>::value == 0 && ka.fmask == 0;
	const bool plain = integer && ka.sflip == 0 && ka.desc == 0;
	const bool hot = (flags & SCATTER_HOT) != 0;
	flags &= ~(u32)SCATTER_HOT;
	// RSX_ELEM_LOADS=1: whole tiles are read with element loads instead of 16-byte loads + a transposition through the LDS
	// (measured on 2^28 u32 keys: 0.490-0.505 against 0.503-0.513 ms per pass in the probe; off by default)
	if (env().elem_loads)
		flags |= SCATTER_ELEM_LOADS;
#define RSX_LAUNCH2(ST, DIGV, HOTV)                                                                                        \
	hipLaunchKernelGGL((rsx_scatter2_kernel<KT, VT, ST, C2, false, DIGV, HOTV, KTO>), grid, dim3(C2::BLOCK), 0, c.stream, kin, kout, \
	                   vin, vout, (u64)n, shift, gbase, tps, (ST *)st, ticket, ka, flags, (u64 *)nullptr, dplan, pass_index,  \
	                   oshift, (const u32 *)c.hotd(), SegArgs{nullptr, c.pass_gate, nullptr, 0, 0, nullptr}, c.pass_alt, c.pass_sp)
	// quarter tiles are for arrays of a few million keys: never 2^30 of them, and hot digits cost little there -- those
	// instantiations are left out of the build
	constexpr bool SMALL_CFG = C2::KPT < Sc2Cfg<KT, VT>::KPT;
#define RSX_LAUNCH2_ST(DIGV)                           \
	do {                                               \
		if constexpr (SMALL_CFG) {                     \
			if (wide)                                  \
				return fail(RSX_EINVAL, "quarter tiles with 2^30 keys or more"); \
			RSX_LAUNCH2(u32, DIGV, false);             \
		} else if (wide) {                             \
			if (hot)                                   \
				RSX_LAUNCH2(u64, DIGV, true);          \
			else                                       \
				RSX_LAUNCH2(u64, DIGV, false);         \
		} else {                                       \
			if (hot)                                   \
				RSX_LAUNCH2(u32, DIGV, true);          \
			else                                       \
				RSX_LAUNCH2(u32, DIGV, false);         \
		}                                              \
	} while (0)
	bool launched = false;
	if constexpr (val_bytes<VT>::value == 0) {
		if (plain) {
			launched = true;
			RSX_LAUNCH2_ST(DIG_PLAIN);
		} else if (integer) {   // signed and / or descending integers: plain digit XOR a per-pass constant
			launched = true;
			RSX_LAUNCH2_ST(DIG_XOR);
		}
	}
	if (!launched)
		RSX_LAUNCH2_ST(DIG_GENERIC);
#undef RSX_LAUNCH2_ST
#undef RSX_LAUNCH2
	HIP_TRY(hipGetLastError());
	if (verify_mode() && dplan && tiles > 0 && !std::is_same<KTO, void>::value && sizeof(KTO) == sizeof(KT)) {
		// a device-scheduled pass (the *_inplace_async sorts): the same check, resolved from the device-side plan like the pass
		// itself; nothing is read back here -- the mismatches add up in a counter of their own until rsx_verify_poll() or the
		// next blocking sort on this stream looks at it
		if (!c.vasync.p) {
			RSX_TRY(c.vasync.ensure(8));
			HIP_TRY(hipMemsetAsync(c.vasync.p, 0, 8, c.stream));
		}
		const u32 vt = (u32)(((u64)(++g_verify_seq) * 2654435761ull) % tiles);
		const u32 kind = (flags & SCATTER_RANK_ASYNC) ? 2u : 1u;
#define RSX_VERIFY_ASYNC(ST)                                                                                            \
		hipLaunchKernelGGL((rsx_verify_tile_kernel<KT, ST>), dim3(1), dim3(64), 0, c.stream, kin, (const void *)kout,         \
		                   (const void *)vin, (const void *)vout, (u64)n, shift, gbase, (const ST *)st, vt, (u32)C2::TILE, ka,  \
		                   (u32)sizeof(KTO), oshift, (u32)val_bytes<VT>::value, 0u, (u64 *)c.vasync.p,                        \
		                   (u32)(env().verify_inject ? 1 : 0), dplan, pass_index, kind, c.pass_alt)
		if (wide)
			RSX_VERIFY_ASYNC(u64);
		else
			RSX_VERIFY_ASYNC(u32);
#undef RSX_VERIFY_ASYNC
		HIP_TRY(hipGetLastError());
	}
	if (verify_mode() && !dplan && tiles > 0) {
		const u32 vt = (u32)(((u64)(++g_verify_seq) * 2654435761ull) % tiles);
		const void *vi = (flags & SCATTER_GEN_INDEX) ? nullptr : (const void *)vin;
#define RSX_VERIFY_TILE(ST)                                                                                             \
		hipLaunchKernelGGL((rsx_verify_tile_kernel<KT, ST>), dim3(1), dim3(64), 0, c.stream, kin, (const void *)kout, vi,       \
		                   (const void *)vout, (u64)n, shift, gbase, (const ST *)st, vt, (u32)C2::TILE, ka, (u32)sizeof(KTO),  \
		                   oshift, (u32)val_bytes<VT>::value, (u32)((flags & SCATTER_SKIP_KEYS) ? 1 : 0), c.verify_bad(),           \
		                   (u32)(env().verify_inject ? 1 : 0))
		if (wide)
			RSX_VERIFY_TILE(u64);
		else
			RSX_VERIFY_TILE(u32);
#undef RSX_VERIFY_TILE
		HIP_TRY(hipGetLastError());
		u64 bad = 0, abad = 0;
		HIP_TRY(hipMemcpyAsync(&bad, c.verify_bad(), sizeof(bad), hipMemcpyDeviceToHost, c.stream));
		if (c.vasync.p)
			HIP_TRY(hipMemcpyAsync(&abad, c.vasync.p, sizeof(abad), hipMemcpyDeviceToHost, c.stream));
		HIP_TRY(hipStreamSynchronize(c.stream));
		if (abad) {
			HIP_TRY(hipMemsetAsync(c.vasync.p, 0, 8, c.stream));
			return fail(RSX_EVERIFY, "RSX_VERIFY: an earlier device-scheduled sort on this stream (rsx_sort*_inplace_async) had a pass "
			                         "whose checked tile differs from its ballot-ranked re-computation in %llu places",
			            (unsigned long long)abad);
		}
		if (bad)
			return fail(RSX_EVERIFY, "RSX_VERIFY: tile %u of a scatter pass (shift %u, %zu keys) differs from its ballot-ranked "
			                         "re-computation in %llu places: the LDS did not return same-address atomics in lane order",
			            vt, shift, n, (unsigned long long)bad);
	}
	return RSX_OK;
}

// quarter tiles? (default tiles for fewer than about a third of the CUs: 10^6 keys: 31 -> 123 tiles, 108 -> 91 us per sort;
// at 10^7 keys, 305 default tiles, quarter tiles are slower: 204 against 178 us)
template <typename KT, typename VT> bool use_small_tiles(size_t n)
{
	if constexpr (Sc2SmallCfg<KT, VT>::AVAILABLE)
		return n < (size_t)96 * Sc2Cfg<KT, VT>::TILE && !env().no_small_tiles;
	return false;
}

// bytes of status words (ticket included) one pass of the fast kernel takes for n elements
template <typename KT, typename VT> size_t status_bytes(size_t n)
{
	const size_t tile = use_small_tiles<KT, VT>(n) ? (size_t)Sc2SmallCfg<KT, VT>::type::TILE : (size_t)Sc2Cfg<KT, VT>::TILE;
	return 256 + (n + tile - 1) / tile * 256 * (n >= (1ull << 30) ? 8 : 4);
}

template <typename KT, typename VT>
int scatter_pass(Ctx &c, const KT *kin, KT *kout, const VT *vin, VT *vout, size_t n, u32 shift, const u64 *gbase,
                 KdfArgs<KT> ka, u32 flags, const Plan *dplan = nullptr, int region = -1, u32 pass_index = 0)
{
	if ((dplan || region >= 0) && !c.fast)
		return fail(RSX_EINVAL, "speculative pass / status regions without the fast kernel");
	if (c.fast) {
		typedef Sc2Cfg<KT, VT> C2;   // count-first kernel (rsx_scatter2.hpp), 32 Ki-key tiles
		typedef Sc2SmallCfg<KT, VT> Small;
		if constexpr (Small::AVAILABLE) {
			if (use_small_tiles<KT, VT>(n))
				return launch_scatter2<KT, VT, typename Small::type>(c, kin, kout, vin, vout, n, shift, gbase, ka, flags, dplan,
				                                                     region, pass_index);
		}
		return launch_scatter2<KT, VT, C2>(c, kin, kout, vin, vout, n, shift, gbase, ka, flags, dplan, region, pass_index);
	}
	typedef ScatterCfg<KT, VT> C1;   // table-ranked fallback (rsx_kernels.hpp)
	flags &= ~(u32)SCATTER_HOT;      // (its match tables do not care how many lanes share a digit)
	const size_t tile = (size_t)C1::TILE;
	const u64 tiles = (n + tile - 1) / tile;
	const u32 tps = choose_tps(n, tile);
	const u64 stiles = (tiles + tps - 1) / tps;
	const bool wide = n >= (1ull << 30);   // counter width by n, as radix_sort.hpp:102-114 does
	const size_t st_bytes = 256 + stiles * 256 * (wide ? 8 : 4);
	RSX_TRY(c.status.ensure(st_bytes));
	HIP_TRY(hipMemsetAsync(c.status.p, 0, st_bytes, c.stream));
	u32 *ticket = (u32 *)c.status.p;
	void *st = (char *)c.status.p + 256;
	ProfScope prof(1, (u64)n * 2 * (sizeof(KT) + val_bytes<VT>::value), c.stream);
	const dim3 grid((unsigned)stiles);
	if (wide)
		hipLaunchKernelGGL((rsx_scatter_kernel<KT, VT, u64>), grid, dim3(C1::BLOCK), 0, c.stream, kin, kout, vin, vout, (u64)n,
		                   shift, gbase, tps, (u64 *)st, ticket, ka, flags, (u64 *)nullptr);
	else
		hipLaunchKernelGGL((rsx_scatter_kernel<KT, VT, u32>), grid, dim3(C1::BLOCK), 0, c.stream, kin, kout, vin, vout, (u64)n,
		                   shift, gbase, tps, (u32 *)st, ticket, ka, flags, (u64 *)nullptr);
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// ---- a rank-sort pass that writes its keys narrowed (fast kernel only; see KTO in rsx_scatter2.hpp) -------------------
template <typename KT, typename VT, typename KTO>
int scatter_pass_narrow(Ctx &c, const KT *kin, KTO *kout, const VT *vin, VT *vout, size_t n, u32 shift, const u64 *gbase,
                        KdfArgs<KT> ka, u32 flags, u32 oshift)
{
	typedef Sc2Cfg<KT, VT> C2;
	typedef Sc2SmallCfg<KT, VT> Small;
	if constexpr (Small::AVAILABLE) {
		if (use_small_tiles<KT, VT>(n))
			return launch_scatter2<KT, VT, typename Small::type, KTO>(c, kin, kout, vin, vout, n, shift, gbase, ka, flags,
			                                                          nullptr, -1, 0, oshift);
	}
	return launch_scatter2<KT, VT, C2, KTO>(c, kin, kout, vin, vout, n, shift, gbase, ka, flags, nullptr, -1, 0, oshift);
}

// keys of `out_bytes` bytes out of a pass over KT keys (out_bytes <= sizeof(KT))
template <typename KT, typename VT>
int scatter_pass_to(Ctx &c, const KT *kin, void *kout, u32 out_bytes, const VT *vin, VT *vout, size_t n, u32 shift,
                    const u64 *gbase, KdfArgs<KT> ka, u32 flags, u32 oshift)
{
	if (out_bytes == sizeof(KT) && oshift == 0)
		return scatter_pass<KT, VT>(c, kin, (KT *)kout, vin, vout, n, shift, gbase, ka, flags);
	if constexpr (sizeof(KT) >= 8)
		if (out_bytes == 8)
			return scatter_pass_narrow<KT, VT, u64>(c, kin, (u64 *)kout, vin, vout, n, shift, gbase, ka, flags, oshift);
	if constexpr (sizeof(KT) >= 4)
		if (out_bytes == 4)
			return scatter_pass_narrow<KT, VT, u32>(c, kin, (u32 *)kout, vin, vout, n, shift, gbase, ka, flags, oshift);
	if constexpr (sizeof(KT) >= 2)
		if (out_bytes == 2)
			return scatter_pass_narrow<KT, VT, uint16_t>(c, kin, (uint16_t *)kout, vin, vout, n, shift, gbase, ka, flags, oshift);
	if (out_bytes == 1)
		return scatter_pass_narrow<KT, VT, uint8_t>(c, kin, (uint8_t *)kout, vin, vout, n, shift, gbase, ka, flags, oshift);
	return fail(RSX_EINVAL, "scatter_pass_to: %u-byte keys out of %zu-byte keys", out_bytes, sizeof(KT));
}

// ---- one MSB pass and leaves (rsx_hybrid.hpp; README.md:647-650) ---------------------------------------------------------
template <typename KT> struct LeafShapes {
	typedef LeafCfg<KT, 4, 32, sizeof(KT) == 8 ? 2 : 4, true, false> Small;   // 8 Ki keys: several workgroups per CU
	typedef LeafCfg<KT, 16, sizeof(KT) == 8 ? 16 : 32> Big;      // as many keys as the LDS stages at once: one workgroup per CU
	// 4-byte keys: a shape in between (16 Ki keys, two workgroups per CU) -- leaves of 8-16 Ki keys (2^29 keys in 65536 buckets)
	// in the large shape were no faster than four passes.  8-byte keys: the small shape already holds 64 KiB.
	static constexpr bool HAS_MEDIUM = sizeof(KT) == 4;
	typedef LeafCfg<KT, 8, 32, 4, true, false> Medium;
	// 4-byte keys, leaves read from slots of at most 5120 keys (2^28 keys in 65536 slots: BASELINE.json's headline): the small
	// shape cut to that size -- twenty rounds per lane instead of thirty-two, 24.5 instead of 37 KiB of LDS: the leaves of 2^28
	// keys take 0.587 instead of 0.640 ms (tools/ubench/leaf_probe, profiles/r03/leaf_probe.txt; with room for a fifth
	// workgroup's registers the compiler spills: 1.5 ms)
	static constexpr bool HAS_FIT = sizeof(KT) == 4;
	typedef LeafCfg<KT, 4, 20, 4, true, false> Fit;
	// ... and two smaller cuts for smaller arrays (the slots of 56 Mi .. 100 Mi keys hold up to 2048 keys, those of up to
	// 157 Mi up to 3072): eight / twelve rounds per lane.  tools/ubench/leaf_probe, 65536 leaves of 1024 keys: 0.229 against
	// 0.346 ms in the 5120-key shape; of 2048 keys: 0.321 against 0.421
	typedef LeafCfg<KT, 4, 8, 8, true, false> Fit2k;
	typedef LeafCfg<KT, 4, 12, 6, true, false> Fit3k;
	// ... the shape (bits 5, 6, 3: the three cuts) for leaves that lie in slots of `cap` keys
	static u32 shape_for_slots(u32 cap)
	{
		if (HAS_FIT && cap <= (u32)Fit2k::CAP)
			return 32u;
		if (HAS_FIT && cap <= (u32)Fit3k::CAP)
			return 64u;
		if (HAS_FIT && cap <= (u32)Fit::CAP)
			return 8u;
		return shape_for(cap);
	}
	// the shape (bit 0 small, bit 2 medium, bit 1 large) for leaves of up to `m` keys
	static u32 shape_for(u32 m)
	{
		if (m <= (u32)Small::CAP)
			return 1u;
		if (HAS_MEDIUM && m <= (u32)Medium::CAP)
			return 4u;
		return 2u;
	}
};

// RSX_NO_HYBRID=1: one pass per kept column whatever the keys look like (the reference's loop, radix_sort.hpp:82-90)
bool hybrid_enabled() { return !env().no_hybrid; }

template <typename KT> HybCaps hybrid_caps(size_t n)
{
	HybCaps caps{0, 0, 0, 0};
	if constexpr (sizeof(KT) >= 4) {
		if (hybrid_enabled() && n < ((size_t)1 << 30)) {
			caps.cap1 = (u32)LeafShapes<KT>::Big::CAP;
			caps.min_cols1 = 3;
			// Two levels pay from about 2^27 keys on (tools/size_sweep.py, profiles/r03/size_sweep.txt: 128 Mi keys 1.16 ms
			// against 1.22 with one pass per column, 256 Mi 1.89 against 2.30; at 64 Mi 0.73 against 0.62 -- a dozen launches
			// and two host round trips are a fixed cost).  Between the reach of one level (about 7 Mi evenly spread keys)
			// and that, one pass per kept column.
			if (n >= ((size_t)1 << env().two_level_min_log2)) {
				caps.cap2 = (u32)LeafShapes<KT>::Big::CAP;   // (leaves beyond the small shape's 8 Ki keys take the large one)
				caps.min_cols2 = 4;
			}
		}
	}
	return caps;
}

// The capacity of a slot for buckets of `mean` keys: 1.25 times the mean, and at least seven standard deviations of an evenly
// spread array's bucket sizes above it, rounded up to 256 keys.  (The second term is what small slots need: with 1.25 x alone a
// mean of 200 keys gets 256-key slots, 3.6 sigma -- evenly spread arrays of 11.5 .. 13 Mi keys overflowed one of their 65536
// slots in one sort out of seven to nine out of ten and were sorted by one pass per column after a lost attempt.)
static inline u32 slot_cap_for(u32 mean)
{
	u32 r = 0;
	while ((u64)(r + 1) * (r + 1) <= mean)
		++r;
	const u32 need = std::max(mean + mean / 4, mean + 7 * (r + 1) + 8);
	return ((need + 255) / 256) * 256;
}

// The capacity -- and the spacing -- of the 256 level-1 slots of a keys-only sort without a histogram.
// The slots fill at the same rate, so the 256 write streams of the level-1 pass stand at the same offset of their slots at any
// time, one slot stride apart: with strides of 15 or 17 x 2 MiB (1.5 x 2^30 four-byte keys: 30 MiB) they meet in the same memory
// channels and the pass runs at 3.6 TB/s instead of 4.5 (tools/stride_probe.py, profiles/r06/stride_probe.txt: +64 KiB .. +1 MiB
// per slot restore it, +4 MiB = 17 x 2 MiB is as bad again).  Slots of a MiB and more are an ODD number of 64 KiB apart
// (RSX_NO_ODD_STRIDE=1: as round 5).  RSX_CAP1_PAD_KIB: that many KiB more per slot (the probe).
template <typename KT> u32 level1_slot_cap(u32 mean)
{
	u32 cap1 = slot_cap_for(mean) + env().cap1_pad_kib * (1024u / (u32)sizeof(KT));
	if (!env().no_odd_stride && (size_t)cap1 * sizeof(KT) >= ((size_t)1 << 20)) {
		const u32 unit = 65536u / (u32)sizeof(KT);
		cap1 = (cap1 + unit - 1) / unit * unit;
		if ((cap1 / unit) % 2u == 0)
			cap1 += unit;
	}
	return cap1;
}

// 8-byte keys: may the sample choose four-byte level-2 slots (SegCtl::narrow)?  Where rsx_leafk_kernel sorts the slots, from
// slots of 512 keys (arrays of ~13 Mi keys) on: the second form of the level-2 pass and of the leaves are two more launches, which
// 8 Mi keys notice (0.267 against 0.252 ms; 16 Mi: 0.328 against 0.337, 64 Mi 0.77 against 0.87, 192 Mi 2.06 against 2.29:
// tools/u64_threshold_probe.py, keys & 0xFFFFFFFFFF).
template <typename KT> bool narrow_slots_ok(u32 cap2)
{
	return sizeof(KT) == 8 && cap2 >= 512u && cap2 <= 5120u && !env().no_leaf16 && !env().no_narrow_slots;
}

// Sorts without a histogram of 4-byte keys (all four columns kept): the level-2 pass writes only the low two bytes of the
// derived keys into its slots and the leaves put the rest back from the slot's digits (RSX_NO_DENSE_SLOTS=1: whole keys).
// (where the slots fit the leaf shape that reads them: up to 5120 keys each, 2^28 keys in all)
// the largest two-byte slot there are leaves for: rsx_leaf16_kernel's 5120 values; round 5: 40960 (slots of 2^31 keys) with the
// counting leaves of rsx_leafc.hpp behind larger shapes of that kernel
constexpr u32 LEAFC_CAP = 40960;
template <typename KT> u32 dense_cap_max()
{
	if (sizeof(KT) != 4)
		return 0u;
	const bool big = !env().no_leafc && !env().no_leaf16 && !env().no_pass16 && !env().no_pass16a && !env().no_unstable;
	return big ? LEAFC_CAP : (u32)LeafShapes<KT>::Fit::CAP;
}
template <typename KT> bool dense_slots(const Ctx &c)
{
	if (sizeof(KT) != 4 || env().no_dense_slots || c.slack_cap == 0 || c.slack_cap > dense_cap_max<KT>())
		return false;
	if (c.slack_cap > (u32)LeafShapes<KT>::Fit::CAP)
		return true;   // (rsx_leafc.hpp: on unless dense_cap_max says otherwise)
	// round 4: rsx_leaf16_kernel (rsx_leaf16.hpp) sorts two-byte slots of every size up to 5120 values faster than the
	// leaves of whole keys are sorted (tools/ubench/leaf16_probe: 2^28 keys 0.39 against 0.67 ms, 2^27 0.25 against 0.46)
	if (!env().no_leaf16)
		return true;
	// RSX_NO_LEAF16=1, round 3's leaves: slots of 3073 .. 5120 keys only (with the smaller cuts the two-byte leaves are level
	// or a little behind -- 64 Mi keys 0.565 against 0.548 ms, 128 Mi 0.867 against 0.862); RSX_DENSE_SLOTS=1: every size
	return env().force_dense_slots || c.slack_cap > (u32)LeafShapes<KT>::Fit3k::CAP;
}

// The leaves of a level (rsx_leaf_sort_kernel).  `shapes`: bit 0 the shape for leaves of up to 8 Ki keys, bit 1 the one that
// fills the LDS; a launched shape does nothing unless the device-side plan has leaves of its size, so both may be enqueued
// before the host knows (nothing then waits for the host).
template <typename KT>
int launch_leaves(Ctx &c, KT *src, KT *aux, size_t n, KdfArgs<KT> ka, u32 level, u32 shapes, const u64 *off1 = nullptr)
{
	typedef typename LeafShapes<KT>::Small S;
	typedef typename LeafShapes<KT>::Big B;
	// one workgroup per bucket at level 1; level 2: a workgroup per table entry (0.569 against 0.585 ms for 2^28 keys with
	// 8192 persistent ones, tools/ubench/leaf_probe; RSX_LEAF_GRID to probe other grids)
	const unsigned grid_s = level == HYB_TWO_LEVEL ? env().leaf_grid : 256u;
	const unsigned grid_1 = 65536u;   // the kernels of rsx_leaf16.hpp take one leaf per workgroup (wave, row): the grid IS the table (LEAF_ONE_PER_GROUP)
	const unsigned grid_b = 256u;
	const LeafSeg *segtab = level == HYB_TWO_LEVEL ? (const LeafSeg *)((char *)c.seg.p + c.seg_segtab_off) : nullptr;
	const SegCtl *ctl = (const SegCtl *)c.seg.p;
	const bool dense = (shapes & 0x100u) != 0;   // (bit 8: the leaves read two-byte slots, dense_slots)
	ProfScope prof(2, (u64)n * (dense ? 2 + sizeof(KT) : 2 * sizeof(KT)), c.stream);
	const KT *slots = level == HYB_TWO_LEVEL ? (const KT *)c.slack.p : nullptr;   // (only leaves of a slack attempt name slots)
	const u32 nopre = env().no_leaf_prefix ? 2u : 0u;   // RSX_NO_LEAF_PREFIX=1: 8-byte-key leaves go through all their columns
	u32 skip_narrowable = nopre;
	if constexpr (sizeof(KT) == 8) {
		if ((shapes & 0x200u) && !env().no_leaf16) {
			// a sort without a histogram, slots of up to 5120 keys: one placement by twelve bits + register passes on 4- or
			// 8-byte values (rsx_leafk_kernel, rsx_leaf16.hpp: the instantiation whose carried type the leaves' columns need
			// works, the other does nothing); what they leave alone goes through the LDS passes of round 3
			u32 *redo = (u32 *)((char *)c.seg.p + c.seg_redo_off);
			SegCtl *wctl = (SegCtl *)c.seg.p;
			// (three shapes by the slots' capacity, as the pairs' leaves: a leaf's fixed costs follow its shape)
#define RSX_LEAFK(K4, K8)                                                                                                      \
	do {                                                                                                                       \
		hipLaunchKernelGGL((rsx_leafk_kernel<KT, u32, K4>), dim3(grid_1), dim3(K4::BLOCK), 0, c.stream, src, aux,               \
		                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)K4::CAP, slots, c.slack_cap, redo,               \
		                   (u32)env().leaf16_maxbin);                                                                          \
		hipLaunchKernelGGL((rsx_leafk_kernel<KT, u64, K8>), dim3(grid_1), dim3(K8::BLOCK), 0, c.stream, src, aux,               \
		                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)K8::CAP, slots, c.slack_cap, redo,               \
		                   (u32)env().leaf16_maxbin);                                                                          \
		if (narrow_slots_ok<KT>(c.slack_cap))   /* four-byte slots (SegCtl::narrow) */                                          \
			hipLaunchKernelGGL((rsx_leafk_kernel<KT, u32, K4, true>), dim3(grid_1), dim3(K4::BLOCK), 0, c.stream, src, aux,     \
			                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)K4::CAP, slots, c.slack_cap, redo,           \
			                   (u32)env().leaf16_maxbin);                                                                      \
	} while (0)
			typedef LeafKCfg<512, 5120, 8> K4;
			typedef LeafKCfg<512, 5120, 8> K8;   // (8-byte values staged in 6 bytes: four workgroups per CU)
			typedef LeafKCfg<256, 2560, 8, 11> K2;
			typedef LeafKCfg<128, 1280, 6, 10> K1;
			typedef LeafKCfg<64, 256, 8, 9> K0;    // slots of up to 256 keys (arrays of up to ~13 Mi keys): a wave per leaf
			typedef LeafKCfg<64, 512, 8, 10> K0b;  // ... and of 512 (arrays of 11.5 .. 27 Mi keys)
			if (c.slack_cap <= (u32)K0::CAP && !env().no_leaf16q)
				RSX_LEAFK(K0, K0);
			else if (c.slack_cap <= (u32)K0b::CAP && !env().no_leaf16q)
				RSX_LEAFK(K0b, K0b);
			else if (c.slack_cap <= (u32)K1::CAP)
				RSX_LEAFK(K1, K1);
			else if (c.slack_cap <= (u32)K2::CAP)
				RSX_LEAFK(K2, K2);
			else {
				// the 5120-key shape (arrays above 2^27 keys: BASELINE.json's cfg 3): the 8-byte-carried leaves by their own kernel
				hipLaunchKernelGGL((rsx_leafk_kernel<KT, u32, K4>), dim3(grid_1), dim3(K4::BLOCK), 0, c.stream, src, aux,
				                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)K4::CAP, slots, c.slack_cap, redo,
				                   (u32)env().leaf16_maxbin);
				hipLaunchKernelGGL((rsx_leafk8_kernel<KT, u64, LeafK8Cfg>), dim3(grid_1), dim3(LeafK8Cfg::BLOCK), 0, c.stream, src, aux,
				                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)LeafK8Cfg::CAP, slots, c.slack_cap, redo,
				                   (u32)env().leaf16_maxbin);
				if (narrow_slots_ok<KT>(c.slack_cap))
					hipLaunchKernelGGL((rsx_leafk_kernel<KT, u32, K4, true>), dim3(grid_1), dim3(K4::BLOCK), 0, c.stream, src, aux,
					                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)K4::CAP, slots, c.slack_cap, redo,
					                   (u32)env().leaf16_maxbin);
			}
#undef RSX_LEAFK
			typedef LeafCfg<u32, 4, 32, 3, true, false> N;
			hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, N, u32>), dim3(2048), dim3(N::BLOCK), 0, c.stream, src, aux, (u64)n,
			                   (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, (u32)S::CAP, slots,
			                   c.slack_cap, nopre, off1, (const u32 *)redo);
			hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, S>), dim3(2048), dim3(S::BLOCK), 0, c.stream, src, aux, (u64)n,
			                   (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, (u32)S::CAP, slots,
			                   c.slack_cap, nopre | 1u, off1, (const u32 *)redo);
			HIP_TRY(hipGetLastError());
			return RSX_OK;
		}
		// 8-byte keys: leaves whose columns all lie in the low four bytes are carried as 4-byte values (rsx_hybrid.hpp, CT)
		if (shapes & 1u) {
			typedef LeafCfg<u32, 4, 32, 3, true, false> N;   // (131 registers: three workgroups per CU)
			static_assert(N::CAP == S::CAP, "the narrow shape takes the small shape's leaves");
			hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, N, u32>), dim3(grid_s), dim3(N::BLOCK), 0, c.stream, src, aux, (u64)n,
			                   (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, (u32)S::CAP, slots,
			                   c.slack_cap, nopre, off1);
			skip_narrowable |= 1u;
		}
	}
	if constexpr (sizeof(KT) == 4) {
		if (dense && !env().no_leaf16) {
			// two-byte slots: one placement by the top bits + two register passes (rsx_leaf16.hpp); what that kernel leaves
			// alone (a list; or everything, if the sample saw the low sixteen bits cluster) goes through the two LDS passes
			u32 *redo = (u32 *)((char *)c.seg.p + c.seg_redo_off);
			SegCtl *wctl = (SegCtl *)c.seg.p;
			if (c.slack_cap > 5120u || (env().force_leafc && c.slack_cap <= LEAFC_CAP)) {
				const unsigned force = env().force_leafc;
				// round 5, arrays beyond 2^28 keys (rsx_leafc.hpp).  Slots of up to 20480 values (2^30 keys): rsx_leaf16_kernel in a larger
				// shape -- 13 or 14 bits name a value's bin, 512 or 1024 threads to a leaf --, and behind it the counting leaves for what
				// it leaves alone; larger slots (2^31 keys: 32 Ki values each): the counting leaves at once.  tools/ubench/leafc_probe,
				// profiles/r05/leafc_probe.txt: 2^29 keys 0.81 ms against 1.78 counting, 2^30 1.80 against 2.35, 2^31 4.33 against 3.20.
				// (the shapes' ladder: tools/ubench/leafc_probe at 300 M, 400 M, 2^29, 700 M, 2^30 and 1.5 x 2^30 keys,
				// profiles/r05/leafc_probe_between.txt -- every step is 10-20 % over the next larger shape at its size)
				typedef Leaf16Cfg<256, 6144, 8, 12> L6k;
				typedef Leaf16Cfg<256, 7680, 8, 12> L7k;
				typedef Leaf16Cfg<512, 10240, 8, 13> L10k;
				typedef Leaf16Cfg<512, 15360, 8, 13> L15k;
				typedef Leaf16Cfg<1024, 20480, 8, 14> L20k;
				const unsigned grid_c = 256u;   // (a workgroup per CU: the cells fill the LDS)
#define RSX_LAUNCH_LC(NVEC, REDO)                                                                                           \
				hipLaunchKernelGGL((rsx_leafc_kernel<KT, LeafCCfg<NVEC>>), dim3(grid_c), dim3(LeafCCfg<NVEC>::BLOCK), 0, c.stream, src, aux, \
				                   (const Plan *)c.plan(), segtab, ctl, ka, 0u, (u32)LeafCCfg<NVEC>::CAP, (const uint16_t *)slots,     \
				                   c.slack_cap, (const u32 *)(REDO))
#define RSX_LAUNCH_L16B(CFG, NVEC)                                                                                          \
				do {                                                                                                        \
					hipLaunchKernelGGL((rsx_leaf16_kernel<KT, CFG>), dim3(grid_1), dim3(CFG::BLOCK), 0, c.stream, src, aux,   \
					                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)CFG::CAP, (const uint16_t *)slots,   \
					                   c.slack_cap, redo, (u32)env().leaf16_maxbin);                                          \
					RSX_LAUNCH_LC(NVEC, redo);                                                                                \
				} while (0)
				const u32 cap = c.slack_cap;
				// (RSX_FORCE_LEAFC, tests: 1 counting, 2 / 3 / 4 / 5 / 6 the 10240- / 20480- / 6144- / 7680- / 15360-value shape)
				const unsigned pick = force ? force
				                            : cap <= (u32)L6k::CAP ? 4u : cap <= (u32)L7k::CAP ? 5u : cap <= (u32)L10k::CAP ? 2u
				                            : cap <= (u32)L15k::CAP ? 6u : cap <= (u32)L20k::CAP ? 3u : 1u;
				if (pick == 4 && cap <= (u32)L6k::CAP)
					RSX_LAUNCH_L16B(L6k, 2);
				else if (pick == 5 && cap <= (u32)L7k::CAP)
					RSX_LAUNCH_L16B(L7k, 2);
				else if (pick == 2 && cap <= (u32)L10k::CAP)
					RSX_LAUNCH_L16B(L10k, 2);
				else if (pick == 6 && cap <= (u32)L15k::CAP)
					RSX_LAUNCH_L16B(L15k, 2);
				else if (pick == 3 && cap <= (u32)L20k::CAP)
					RSX_LAUNCH_L16B(L20k, 3);
				else if (cap <= (u32)LeafCCfg<4>::CAP)
					RSX_LAUNCH_LC(4, nullptr);
				else
					RSX_LAUNCH_LC(5, nullptr);
#undef RSX_LAUNCH_L16B
#undef RSX_LAUNCH_LC
				HIP_TRY(hipGetLastError());
				return RSX_OK;
			}
			typedef Leaf16Cfg<256, 5120, 8, 12> L5k;
			// (128 threads per leaf for slots of up to 2560 values -- arrays of 52 Mi .. 128 Mi keys: a 1280-value leaf keeps 80 lanes
			// busy in the register passes, and sixteen small workgroups per CU overlap better than eight: tools/ubench/leaf16_probe,
			// profiles/r05/leaf16_probe_mid.txt: 56 Mi keys 0.118 against 0.163 ms, 128 Mi 0.214 against 0.252)
			typedef Leaf16Cfg<128, 2560, 8, 11> L2k;
			typedef Leaf16WCfg<1024, 10, 4> W1k;   // small slots (arrays of up to ~50 Mi keys): a wave per leaf
			typedef Leaf16WCfg<512, 9, 4> W512;
			// ... and, round 5, up to 2048 values (arrays of up to ~100 Mi keys: two chunks of sixteen values per lane in the register
			// passes, slots read from both ends behind rsx_pass16a_kernel): tools/ubench/leaf16_probe against the 128-thread
			// workgroup shape -- 54 Mi keys 0.110 against 0.119 ms, 64 Mi 0.131 / 0.143, 80 Mi 0.145 / 0.167, 96 Mi 0.162 / 0.180
			typedef Leaf16WCfg<2048, 10, 4> W2k;
			if (c.slack_cap > (u32)W1k::CAP && c.slack_cap <= (u32)W2k::CAP && !env().no_leaf16w2k) {
				hipLaunchKernelGGL((rsx_leaf16w_kernel<KT, W2k>), dim3(grid_1 / W2k::NW), dim3(W2k::BLOCK), 0, c.stream, src, aux,
				                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)W2k::CAP, (const uint16_t *)slots, c.slack_cap);
				HIP_TRY(hipGetLastError());
				return RSX_OK;
			}
			if (c.slack_cap <= (u32)W1k::CAP) {
				// (no list, no second launch: the wave kernel goes on until its leaf is in order)
				typedef Leaf16QCfg<4> Q256;            // slots of up to 256 values (arrays of up to ~13 Mi keys): four leaves per wave
				if (c.slack_cap <= (u32)Q256::CAP && !env().no_leaf16q)
					hipLaunchKernelGGL((rsx_leaf16q_kernel<KT, Q256>), dim3(grid_1 / Q256::ROWS), dim3(Q256::BLOCK), 0, c.stream, src, aux,
					                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)Q256::CAP, (const uint16_t *)slots, c.slack_cap);
				else if (c.slack_cap <= (u32)W512::CAP)
					hipLaunchKernelGGL((rsx_leaf16w_kernel<KT, W512>), dim3(grid_1 / W512::NW), dim3(W512::BLOCK), 0, c.stream, src, aux,
					                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)W512::CAP, (const uint16_t *)slots, c.slack_cap);
				else
					hipLaunchKernelGGL((rsx_leaf16w_kernel<KT, W1k>), dim3(grid_1 / W1k::NW), dim3(W1k::BLOCK), 0, c.stream, src, aux,
					                   (const Plan *)c.plan(), segtab, wctl, ka, 0u, (u32)W1k::CAP, (const uint16_t *)slots, c.slack_cap);
				HIP_TRY(hipGetLastError());
				return RSX_OK;
			}
#define RSX_LAUNCH_L16(KERNEL, CFG, GRID)                                                                                     \
			hipLaunchKernelGGL((KERNEL<KT, CFG>), dim3(GRID), dim3(CFG::BLOCK), 0, c.stream, src, aux, (const Plan *)c.plan(),  \
			                   segtab, wctl, ka, 0u, (u32)CFG::CAP, (const uint16_t *)slots, c.slack_cap, redo,                \
			                   (u32)env().leaf16_maxbin)
			if (c.slack_cap <= (u32)L2k::CAP)
				RSX_LAUNCH_L16(rsx_leaf16_kernel, L2k, grid_1);
			else
				RSX_LAUNCH_L16(rsx_leaf16_kernel, L5k, grid_1);
#undef RSX_LAUNCH_L16
			typedef typename LeafShapes<KT>::Fit F_;
			hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, F_, uint16_t, true>), dim3(4096), dim3(F_::BLOCK), 0, c.stream, src, aux,
			                   (u64)n, (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, (u32)F_::CAP, slots,
			                   c.slack_cap, nopre, off1, (const u32 *)redo);
			HIP_TRY(hipGetLastError());
			return RSX_OK;
		}
	}
	if constexpr (LeafShapes<KT>::HAS_FIT) {
		// the shapes cut to the slots' size (exactly one of them is asked for; each takes the leaves up to its capacity)
#define RSX_LAUNCH_FIT(BIT, SHAPE)                                                                                          \
		if (shapes & (BIT)) {                                                                                               \
			typedef typename LeafShapes<KT>::SHAPE F_;                                                                      \
			if (dense)                                                                                                      \
				hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, F_, uint16_t, true>), dim3(grid_s), dim3(F_::BLOCK), 0, c.stream, \
				                   src, aux, (u64)n, (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, \
				                   (u32)F_::CAP, slots, c.slack_cap, nopre, off1);                                          \
			else                                                                                                            \
				hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, F_>), dim3(grid_s), dim3(F_::BLOCK), 0, c.stream, src, aux,     \
				                   (u64)n, (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u,       \
				                   (u32)F_::CAP, slots, c.slack_cap, nopre, off1);                                          \
		}
		RSX_LAUNCH_FIT(32u, Fit2k)
		RSX_LAUNCH_FIT(64u, Fit3k)
		RSX_LAUNCH_FIT(8u, Fit)
#undef RSX_LAUNCH_FIT
	}
	if (shapes & 1u)
		hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, S>), dim3(grid_s), dim3(S::BLOCK), 0, c.stream, src, aux, (u64)n,
		                   (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, 0u, (u32)S::CAP, slots,
		                   c.slack_cap, skip_narrowable, off1);
	typedef typename LeafShapes<KT>::Medium M;
	const u32 big_lo = LeafShapes<KT>::HAS_MEDIUM ? (u32)M::CAP : (u32)S::CAP;
	if constexpr (LeafShapes<KT>::HAS_MEDIUM) {
		if (shapes & 4u)
			hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, M>), dim3(level == HYB_TWO_LEVEL ? env().leaf_grid : 256u), dim3(M::BLOCK), 0, c.stream, src,
			                   aux, (u64)n, (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, (u32)S::CAP, (u32)M::CAP,
			                   slots, c.slack_cap, nopre, off1);
	}
	if (shapes & 2u)
		hipLaunchKernelGGL((rsx_leaf_sort_kernel<KT, B>), dim3(grid_b), dim3(B::BLOCK), 0, c.stream, src, aux, (u64)n,
		                   (const u64 *)c.ghist(), (const Plan *)c.plan(), segtab, ctl, ka, level, big_lo, (u32)B::CAP, slots,
		                   c.slack_cap, nopre, off1);
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// The level-2 pass of a keys-only sort of 4-byte keys without a histogram writes whole 64-byte atoms (rsx_pass16a_kernel,
// rsx_pass16.hpp) where the leaves are rsx_leaf16_kernel's (slots of more than 1024 values: arrays from about 52 Mi keys),
// which read a slot from both ends; its tiles are Pass16aCfg::TILE keys.
template <typename KT> bool pass16a_wanted(const Ctx &c)
{
	if constexpr (sizeof(KT) == 4)
	{
		if (!dense_slots<KT>(c) || env().no_pass16 || env().no_pass16a || env().no_unstable || env().no_leaf16)
			return false;
		if (c.slack_cap > 1024u)
			return true;
		// slots of 1024 values (38 .. 52 M keys; the wave per leaf reads both ends since round 5): where the 128 places kept for a
		// slot's back still leave its front the room slot_cap_for wanted for the whole slot -- mean + 7 standard deviations:
		// 37.7 M .. 45.9 M keys (the reference's own headline size, 4 * 10^7, among them)
		if (c.slack_cap == 1024u && c.slack_mean) {
			u32 r = 0;
			while ((u64)(r + 1) * (r + 1) <= c.slack_mean)
				++r;
			return c.slack_mean + 7 * (r + 1) + 8 <= c.slack_cap - LEAF16_BACK;
		}
		return false;
	}
	return false;
}

// a pass inside the level-1 buckets (SEG instantiation of the pass kernel): j < 0 the one by the level-2 column (runs in
// SEG_MODE_LEAVES), j >= 0 LSB-first pass j (runs in SEG_MODE_LSD).  aux -> src, src -> aux for odd j.
// j == -2: the slack attempt (aux -> the slots of c.slack, no counts needed).
// blind (a sort without a histogram, sort_keys_blind): 1 = its level-1 pass (`aux` = the caller's array -> the 256 slots of
// c.slack1, by the top column, status region 1), 2 = its level-2 pass (j == -2, reading c.slack1 instead of `aux`).
// Rows of status words (and tile-table entries) a segmented pass may need beyond n / TILE: a partial tile per level-1 bucket, and
// -- 8-byte keys, whose level-1 pass may be rsx_pass32a_kernel in front of the CHAINED level-2 pass -- one more per bucket for
// what lies at its slot's end (rsx_seg_tiles_kernel, back_cap).
template <typename KT> constexpr u64 seg_extra_rows() { return sizeof(KT) == 8 ? 512 : 256; }

// 8-byte keys, the level-2 pass into FOUR-byte slots (SegCtl::narrow) as rsx_pass64a_kernel: whole atoms, cursors, two-ended slots
// (rsx_leafk_kernel's SLOT32 form reads both ends whatever its shape)
template <typename KT> bool pass64a_narrow_wanted(const Ctx &c)
{
	if (sizeof(KT) != 8 || !narrow_slots_ok<KT>(c.slack_cap) || env().no_pass64a || env().no_unstable || !c.slack_mean)
		return false;
	// the 128 places kept for a slot's back must leave its front mean + 6 standard deviations (what is carried to the back, a
	// few dozen values per slot, comes on top): just below a step of slot_cap_for they do not -- 64 Mi keys, mean 1024 in slots
	// of 1280, lost the attempt -- and the chained pass, whose slots have no back, stays
	u32 r = 0;
	while ((u64)(r + 1) * (r + 1) <= c.slack_mean)
		++r;
	return c.slack_mean + 6 * (r + 1) + 8 <= c.slack_cap - Pass2wCfg<u32>::BACK;
}

template <typename KT>
int launch_seg_pass(Ctx &c, const KT *aux, KT *src, size_t n, KdfArgs<KT> ka, int j, int blind = 0)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	const u64 rows = (n + C2::TILE - 1) / C2::TILE + seg_extra_rows<KT>();
	const size_t st_bytes = 256 + rows * 256 * 4;
	char *base = (char *)c.seg.p + c.seg_status_off + (size_t)(blind == 1 ? 1 : j < 0 ? 0 : j) * st_bytes;
	SegArgs sa{};
	sa.ctl = (const SegCtl *)c.seg.p;
	sa.hist = (const u32 *)((char *)c.seg.p + c.seg_hist_off);
	sa.tiles = (const SegTile *)((char *)c.seg.p + c.seg_tiles_off);
	sa.slots = (u32)sizeof(KT) - 1;
	sa.slack_cap = blind == 1 ? c.slack1_cap : j == -2 ? c.slack_cap : 0u;
	sa.overflow = &((SegCtl *)c.seg.p)->overflow;
	KT *const second = blind == 1 ? src : blind == 2 ? const_cast<KT *>(aux) : nullptr;
	if (j == -2)
		src = (KT *)c.slack.p;
	// blind: `src` (level-1 pass) / `aux` (level-2 pass) name the caller's second buffer when the first slack1_lo level-1 slots
	// lie there (blind_enqueue); the others lie in c.slack1
	if (blind == 1 || blind == 2) {
		KT *first = (KT *)c.slack1.p;
		if (second && c.slack1_lo) {
			sa.lo_slots = c.slack1_lo;
			// (virtual slot 0 of the scratch part: slack1_lo slots before the array)
			const uintptr_t lo_a = (uintptr_t)second, hi_a = (uintptr_t)c.slack1.p - (size_t)c.slack1_lo * c.slack1_cap * sizeof(KT);
			if (blind == 1) {
				// one base for the level-1 pass's stores, the parts' offsets in its run offsets (blind_enqueue has checked
				// that both lie within 2^32 elements of the lower one)
				const uintptr_t base_a = std::min(lo_a, hi_a);
				sa.out_off_lo = (u32)((lo_a - base_a) / sizeof(KT));
				sa.out_off_hi = (u32)((hi_a - base_a) / sizeof(KT));
				first = (KT *)base_a;
			} else {
				sa.kin_hi = (const void *)hi_a;
				first = second;
			}
		}
		if (blind == 1)
			src = first;
		else
			aux = first;
	}
	const bool dense = sizeof(KT) == 4 && blind == 2 && dense_slots<KT>(c);   // (keys written as two bytes: its own line in the profile)
	ProfScope prof(dense ? 3 : 1, (u64)n * (dense ? sizeof(KT) + 2 : 2 * sizeof(KT)), c.stream);
	const bool plain = ka.fmask == 0 && ka.sflip == 0 && ka.desc == 0;
	u32 flags = j == -2 ? (u32)SCATTER_SEG_SLACK : j < 0 ? (u32)SCATTER_SEG_LEAVES : 0u;
	if (blind)
		flags |= SCATTER_BLIND | (blind == 1 ? (u32)SCATTER_BLIND_TOP : 0u);
	// keys only, and what these two passes write is sorted by leaves that do not care in which order a bucket's keys arrive
	// (any ascending order of equal bits is the reference's output): no row of cells per wave, no layout over the rows
	if (blind && !env().no_unstable)
		flags |= SCATTER_UNSTABLE;
	const u32 pi = j < 0 ? 0u : (u32)j;
	const unsigned grid = blind == 1 ? (unsigned)(rows - seg_extra_rows<KT>()) : (unsigned)rows;
	const u32 shift0 = 0u;   // (every segmented pass reads its column from the device-side plan)
#define RSX_LAUNCH_SEG(DIGV)                                                                                               \
	hipLaunchKernelGGL((rsx_scatter2_kernel<KT, NoVal, u32, C2, false, DIGV, false, KT, true>), dim3(grid),                 \
	                   dim3(C2::BLOCK), 0, c.stream, aux, src, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, shift0,     \
	                   (const u64 *)c.ghist(), 1u, (u32 *)(base + 256), (u32 *)base, ka, flags, (u64 *)nullptr,             \
	                   (const Plan *)c.plan(), pi, 0u, (const u32 *)nullptr, sa)
	if constexpr (sizeof(KT) == 4) {
		if (dense && pass16a_wanted<KT>(c)) {
			// ... and with whole 64-byte atoms: a workgroup takes a range of tiles and carries what does not fill an atom
			const unsigned pgrid = 512;
			if (plain)
				hipLaunchKernelGGL((rsx_pass16a_kernel<KT, DIG_PLAIN>), dim3(pgrid), dim3(Pass16aCfg::BLOCK), 0, c.stream, (const KT *)aux,
				                   (const KT *)sa.kin_hi, sa.lo_slots, (unsigned short *)src, sa.tiles, sa.ctl, (const Plan *)c.plan(),
				                   (u32 *)(base + 256), sa.slack_cap, sa.overflow, ka);
			else
				hipLaunchKernelGGL((rsx_pass16a_kernel<KT, DIG_GENERIC>), dim3(pgrid), dim3(Pass16aCfg::BLOCK), 0, c.stream, (const KT *)aux,
				                   (const KT *)sa.kin_hi, sa.lo_slots, (unsigned short *)src, sa.tiles, sa.ctl, (const Plan *)c.plan(),
				                   (u32 *)(base + 256), sa.slack_cap, sa.overflow, ka);
			HIP_TRY(hipGetLastError());
			return RSX_OK;
		}
		if (dense && !env().no_pass16 && !env().no_unstable && !env().no_leaf16) {
			// round 5: the pass as a kernel of its own (rsx_pass16.hpp): values staged in two bytes, two workgroups per CU, cursors
			// instead of the chain, 16-byte stores.  (Its slots hold a bucket's values in arbitrary order: for leaves that sort.)
			const u32 *btile = (const u32 *)((char *)c.seg.p + c.seg_btile_off);
#define RSX_LAUNCH_P16(DIGV, CFG)                                                                                          \
	hipLaunchKernelGGL((rsx_pass16_kernel<KT, DIGV, CFG>), dim3(grid), dim3(CFG::BLOCK), 0, c.stream, (const KT *)aux,      \
	                   (const KT *)sa.kin_hi, sa.lo_slots, (unsigned short *)src, sa.tiles, btile, sa.ctl,                  \
	                   (const Plan *)c.plan(), (u32 *)(base + 256), sa.slack_cap, sa.overflow, ka, (u32)env().pass16_dbg)
			if (env().pass16_wgs == 1) {
				if (plain)
					RSX_LAUNCH_P16(DIG_PLAIN, Pass16Cfg<1>);
				else
					RSX_LAUNCH_P16(DIG_GENERIC, Pass16Cfg<1>);
			} else {
				if (plain)
					RSX_LAUNCH_P16(DIG_PLAIN, Pass16Cfg<2>);
				else
					RSX_LAUNCH_P16(DIG_GENERIC, Pass16Cfg<2>);
			}
#undef RSX_LAUNCH_P16
			HIP_TRY(hipGetLastError());
			return RSX_OK;
		}
		if (dense) {
			// 4-byte keys, every column kept: the leaves sort by the two low bytes and the slot says the rest -- the pass writes
			// the low half of every DERIVED key (rsx_leaf_sort_kernel, DENSE)
#define RSX_LAUNCH_SEG16(DIGV)                                                                                             \
	hipLaunchKernelGGL((rsx_scatter2_kernel<KT, NoVal, u32, C2, false, DIGV, false, uint16_t, true>), dim3(grid),           \
	                   dim3(C2::BLOCK), 0, c.stream, aux, (uint16_t *)src, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n, \
	                   shift0, (const u64 *)c.ghist(), 1u, (u32 *)(base + 256), (u32 *)base, ka, flags, (u64 *)nullptr,      \
	                   (const Plan *)c.plan(), pi, 0u, (const u32 *)nullptr, sa)
			if (plain)
				RSX_LAUNCH_SEG16(DIG_PLAIN);
			else
				RSX_LAUNCH_SEG16(DIG_GENERIC);
#undef RSX_LAUNCH_SEG16
			HIP_TRY(hipGetLastError());
			return RSX_OK;
		}
	}
	if (plain)
		RSX_LAUNCH_SEG(DIG_PLAIN);
	else
		RSX_LAUNCH_SEG(DIG_GENERIC);
#undef RSX_LAUNCH_SEG
	if constexpr (sizeof(KT) == 8) {
		if (blind == 2 && narrow_slots_ok<KT>(c.slack_cap)) {
			// ... and the form that writes the low word of every derived key (SegCtl::narrow decides on the device which of the
			// two works; it uses its own status words: the same region, which the form that left has not touched)
#define RSX_LAUNCH_SEG32(DIGV)                                                                                             \
	hipLaunchKernelGGL((rsx_scatter2_kernel<KT, NoVal, u32, C2, false, DIGV, false, u32, true>), dim3(grid),                \
	                   dim3(C2::BLOCK), 0, c.stream, aux, (u32 *)src, (const NoVal *)nullptr, (NoVal *)nullptr, (u64)n,     \
	                   shift0, (const u64 *)c.ghist(), 1u, (u32 *)(base + 256), (u32 *)base, ka, flags, (u64 *)nullptr,     \
	                   (const Plan *)c.plan(), pi, 0u, (const u32 *)nullptr, sa)
			if (pass64a_narrow_wanted<KT>(c)) {
				typedef Pass2wCfg<u32> P64;
				hipLaunchKernelGGL((rsx_pass64a_kernel<KT, u32>), dim3(P64::GRID), dim3(P64::BLOCK), 0, c.stream, (const KT *)aux,
				                   (const KT *)sa.kin_hi, sa.lo_slots, (u32 *)src, sa.tiles, sa.ctl, (const Plan *)c.plan(),
				                   (u32 *)(base + 256), sa.slack_cap, sa.overflow, ka);
				if (c.narrow1 && second) {
					// SegCtl::narrow == 2: the level-1 slots are four-byte places in the caller's second buffer (blind_enqueue), the
					// same element indices; what they hold is derived already
					typedef Pass64aCfgLow P64L;
					hipLaunchKernelGGL((rsx_pass64a_kernel<u32, u32, P64L>), dim3(P64L::GRID), dim3(P64L::BLOCK), 0, c.stream, (const u32 *)second,
					                   (const u32 *)nullptr, 0u, (u32 *)src, sa.tiles, sa.ctl, (const Plan *)c.plan(),
					                   (u32 *)(base + 256), sa.slack_cap, sa.overflow, KdfArgs<u32>{0, 0, 0});
				}
			} else if (plain)
				RSX_LAUNCH_SEG32(DIG_PLAIN);
			else
				RSX_LAUNCH_SEG32(DIG_GENERIC);
#undef RSX_LAUNCH_SEG32
		}
	}
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// where the parts of a two-level sort's device-side state lie in c.seg
// bytes of c.seg for a two-level sort of n keys (seg_layout below)
// rows of the tile table of a two-level sort's level-2 pass: the tiles of the pass that may run (the chained pass's, or the smaller
// ones of rsx_pass16a_kernel / rsx_pass64a_kernel) + per bucket a partial tile and one for what lies at its slot's end
template <typename KT> u64 seg_tile_rows(size_t n, u64 rows)
{
	if (sizeof(KT) == 4)
		return (n + Pass16aCfg::TILE - 1) / Pass16aCfg::TILE + 514;
	return std::max<u64>(rows, (n + Pass2wCfg<u32>::TILE - 1) / Pass2wCfg<u32>::TILE + 514);
}

template <typename KT> size_t seg_bytes(size_t n)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	const u64 rows = (n + C2::TILE - 1) / C2::TILE + seg_extra_rows<KT>();
	const size_t st_bytes = 256 + rows * 256 * 4;
	const size_t hist_bytes = (size_t)256 * (sizeof(KT) - 1) * 256 * sizeof(u32);
	const u64 tile_rows = seg_tile_rows<KT>(n, rows);
	return 256 + hist_bytes + (sizeof(KT) - 1) * st_bytes + 65536 * sizeof(LeafSeg) + tile_rows * sizeof(SegTile) + 260 * sizeof(u32) +
	       65536 * sizeof(u32);
}

template <typename KT> int seg_layout(Ctx &c, size_t n)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	const u64 rows = (n + C2::TILE - 1) / C2::TILE + seg_extra_rows<KT>();
	const size_t st_bytes = 256 + rows * 256 * 4;
	const size_t hist_bytes = (size_t)256 * (sizeof(KT) - 1) * 256 * sizeof(u32);
	c.seg_hist_off = 256;
	c.seg_status_off = c.seg_hist_off + hist_bytes;
	c.seg_segtab_off = c.seg_status_off + (sizeof(KT) - 1) * st_bytes;
	c.seg_tiles_off = c.seg_segtab_off + 65536 * sizeof(LeafSeg);
	const u64 tile_rows = seg_tile_rows<KT>(n, rows);
	c.seg_btile_off = c.seg_tiles_off + tile_rows * sizeof(SegTile);
	c.seg_redo_off = c.seg_btile_off + 260 * sizeof(u32);   // the leaves rsx_leaf16_kernel leaves to rsx_leaf_sort_kernel
	const void *before = c.seg.p;
	RSX_TRY(c.seg.ensure(c.seg_redo_off + 65536 * sizeof(u32)));
	if (c.seg.p != before || c.seg.external)   // (a new control block -- or a caller's workspace, whose contents are scratch: SegCtl::boff_*)
		HIP_TRY(hipMemsetAsync(c.seg.p, 0, 256, c.stream));
	return RSX_OK;
}

// What a sort of n keys WITHOUT a histogram needs on top of that (blind_enqueue): the level-1 slots that do not fit the
// caller's second buffer, the level-2 slots, the control block / tables / status words of the two passes.
template <typename KT> void blind_sizes(size_t n, size_t *gscan, size_t *seg, size_t *slack1, size_t *slack)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	const u32 cap1 = level1_slot_cap<KT>((u32)(n >> 8)), cap2 = slot_cap_for((u32)(n >> 16));
	const u32 lo = cap1 >= (u32)C2::TILE ? (u32)std::min<size_t>(n / cap1, 255) : 0u;
	const size_t slot2 = (sizeof(KT) == 4 && cap2 <= dense_cap_max<KT>()) ? 2 : sizeof(KT);
	*gscan = 256 * sizeof(u64);
	*seg = (seg_bytes<KT>(n) + 255) & ~(size_t)255;
	*slack1 = ((((size_t)(256 - lo) * cap1 + C2::TILE) * sizeof(KT)) + 255) & ~(size_t)255;
	*slack = ((((size_t)65536 * cap2 + C2::TILE) * slot2) + 255) & ~(size_t)255;
}

// ... handed to the context if the workspace has it (rsx_workspace_bytes_fast): the attempt is then made inside the workspace
template <typename KT> void borrow_blind(Ctx &v, char *p, char *ws_end, size_t n)
{
	if constexpr (sizeof(KT) >= 4) {
		size_t g, sg, s1, s2;
		blind_sizes<KT>(n, &g, &sg, &s1, &s2);
		p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
		if (n < ((size_t)1 << 22) || n >= ((size_t)1 << 30) || p + g + sg + s1 + s2 > ws_end)
			return;
		v.gscan.borrow(p, g);
		p += g;
		v.seg.borrow(p, sg);
		p += sg;
		v.slack1.borrow(p, s1);
		p += s1;
		v.slack.borrow(p, s2);
		v.ws_blind = true;
	}
}


// The second level of a two-level sort.  Pass 1 (by the highest kept column, src -> aux) is on its way; `plan` says so.
// Ends with the sorted keys in the buffer the reference's parity rule names (radix_sort.hpp:92); *result says which.
template <typename KT>
int sort_keys_two_level(Ctx &c, KT *src, KT *aux, size_t n, KdfArgs<KT> ka, const Plan &plan, KT **result, u32 *how)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	const u64 rows = (n + C2::TILE - 1) / C2::TILE + seg_extra_rows<KT>();
	const size_t st_bytes = 256 + rows * 256 * 4;
	RSX_TRY(seg_layout<KT>(c, n));
	SegCtl *ctl = (SegCtl *)c.seg.p;
	u32 *seghist = (u32 *)((char *)c.seg.p + c.seg_hist_off);
	SegTile *tiles = (SegTile *)((char *)c.seg.p + c.seg_tiles_off);
	LeafSeg *segtab = (LeafSeg *)((char *)c.seg.p + c.seg_segtab_off);
	u32 *btile = (u32 *)((char *)c.seg.p + c.seg_btile_off);
	KT *final = (plan.ncols & 1) ? aux : src;
	if (!c.seg_ev)
		HIP_TRY(hipEventCreateWithFlags(&c.seg_ev, hipEventDisableTiming));
	// control block, digit counts and the status words of the first segmented pass, zeroed together
	HIP_TRY(hipMemsetAsync(c.seg.p, 0, c.seg_status_off + st_bytes, c.stream));
	hipLaunchKernelGGL(rsx_seg_tiles_kernel, dim3(32), dim3(256), 0, c.stream, (const u64 *)c.ghist(), (u64)n, (const Plan *)c.plan(),
	                   (u32)C2::TILE, tiles, ctl, btile);
	HIP_TRY(hipGetLastError());
	// The slack attempt: evenly spread keys need no counts for the second pass.  Every (digit, digit) bucket gets a slot of
	// 1.25 times its expected size in a scratch array and the pass writes each key where the look-back chain puts it inside
	// its bucket's slot; the bucket sizes are then read off the chain, and the leaves gather from the slots into the dense
	// result.  One read of the keys less than the counted path below (rsx_seg_hist1_kernel: 0.25 of 2.1 ms at 2^28 keys).
	// A slot that overflows (keys clustered after all) only costs the attempt: pass 1's output in `aux` is untouched.
	c.slack_cap = 0;
	if (!env().no_slack && n >= ((size_t)1 << 26)) {
		const u32 mean = (u32)(n >> 16);
		const u32 cap = slot_cap_for(mean);
		if (cap <= (u32)LeafShapes<KT>::Big::CAP && c.slack.ensure(((size_t)65536 * cap + C2::TILE) * sizeof(KT)) == RSX_OK) {
			c.slack_cap = cap;
			RSX_TRY(launch_seg_pass<KT>(c, aux, src, n, ka, -2));
			hipLaunchKernelGGL((rsx_seg_slack_plan_kernel<u32>), dim3(256), dim3(256), 0, c.stream,
			                   (const u32 *)((char *)c.seg.p + c.seg_status_off + 256), (const u32 *)btile, (const u64 *)c.ghist(),
			                   (const Plan *)c.plan(), ctl, segtab, cap, c.dev_host_segctl);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipEventRecord(c.seg_ev, c.stream));
			RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_TWO_LEVEL, LeafShapes<KT>::shape_for_slots(cap)));
			HIP_TRY(hipEventSynchronize(c.seg_ev));
			if (c.host_segctl->mode == SEG_MODE_LEAVES) {
				*result = final;
				*how = 4u;
				return RSX_OK;
			}
			// a slot overflowed: the counted path, from `aux` again
			c.slack_cap = 0;
			HIP_TRY(hipMemsetAsync(c.seg.p, 0, c.seg_status_off + st_bytes, c.stream));
			hipLaunchKernelGGL(rsx_seg_tiles_kernel, dim3(32), dim3(256), 0, c.stream, (const u64 *)c.ghist(), (u64)n,
			                   (const Plan *)c.plan(), (u32)C2::TILE, tiles, ctl, btile);
		} else {
			(void)hipGetLastError();
		}
	}
	{
		ProfScope prof(0, (u64)n * sizeof(KT), c.stream);
		hipLaunchKernelGGL((rsx_seg_hist1_kernel<KT>), dim3(512), dim3(1024), 0, c.stream, (const KT *)aux, (const SegTile *)tiles,
		                   (const SegCtl *)ctl, (const Plan *)c.plan(), ka, seghist);
	}
	hipLaunchKernelGGL((rsx_seg_plan_kernel<KT>), dim3(256), dim3(256), 0, c.stream, seghist, (const u64 *)c.ghist(), (u64)n,
	                   (const Plan *)c.plan(), ctl, segtab, (u32)LeafShapes<KT>::Big::CAP, c.dev_host_segctl, 0u);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c.seg_ev, c.stream));
	// the pass by the level-2 column and the small leaves are enqueued before the host knows whether the (digit, digit)
	// buckets fit leaves: they do nothing if not
	RSX_TRY(launch_seg_pass<KT>(c, aux, src, n, ka, -1));
	RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_TWO_LEVEL, 1u));
	HIP_TRY(hipEventSynchronize(c.seg_ev));
	const SegCtl hc = *c.host_segctl;
	if (hc.mode == SEG_MODE_LEAVES) {
		if (hc.maxleaf > (u32)LeafShapes<KT>::Small::CAP)   // (rare: the leaves need a larger shape)
			RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_TWO_LEVEL, LeafShapes<KT>::shape_for(hc.maxleaf)));
	} else {
		// keys clustered in their top two columns: one pass per remaining column inside the level-1 buckets, LSB first
		// (the counts of the columns below the level-2 one are only made now)
		if (plan.ncols > 2) {
			ProfScope prof(0, (u64)n * sizeof(KT), c.stream);
			hipLaunchKernelGGL((rsx_seg_hist_kernel<KT>), dim3(512), dim3(1024), 0, c.stream, (const KT *)aux, (const SegTile *)tiles,
			                   (const SegCtl *)ctl, (const Plan *)c.plan(), ka, seghist);
		}
		hipLaunchKernelGGL((rsx_seg_plan_kernel<KT>), dim3(256), dim3(256), 0, c.stream, seghist, (const u64 *)c.ghist(), (u64)n,
		                   (const Plan *)c.plan(), ctl, segtab, 0u, (SegCtl *)nullptr, 1u);
		if (plan.ncols > 2)
			HIP_TRY(hipMemsetAsync((char *)c.seg.p + c.seg_status_off + st_bytes, 0, (plan.ncols - 2) * st_bytes, c.stream));
		for (u32 j = 0; j + 1 < plan.ncols; ++j)
			RSX_TRY(launch_seg_pass<KT>(c, aux, src, n, ka, (int)j));
	}
	*result = final;
	*how = hc.mode == SEG_MODE_LEAVES ? 2u : 3u;
	return RSX_OK;
}

// ---- two MSB passes and leaves WITHOUT the histogram (rsx_hybrid.hpp, rsx_blind_precheck_kernel) -----------------------------
// For the arrays a two-level sort is for (hybrid_caps: cap2), blocking keys-only sorts.  *done = 1: sorted, *result set.
// *done = 0: called off (the sample did not prove what it has to, or a slot overflowed) -- `src` and `aux` are untouched and
// the caller runs the ordinary path.  A context that has been called off skips the next attempts of its kind (1, 3, 7 ... 31 sorts).
template <typename KT> HybCaps hybrid_caps_pairs(size_t n, size_t val_bytes_);
// kinds of sorts that learn separately: 0 / 1 keys only (4- / 8-byte keys), 2 rank sorts, 3 key + payload sorts
template <typename KT> constexpr int blind_kind(size_t payload_bytes, bool rank = false)
{
	return payload_bytes ? (rank ? 2 : 3) : (sizeof(KT) == 8 ? 1 : 0);
}
inline void blind_called_off(Ctx &c, int kind)
{
	c.blind_backoff[kind] = std::min<u32>(2 * c.blind_backoff[kind] + 1, 31);
	c.blind_skip[kind] = c.blind_backoff[kind];
}
// rsx_reload_env() makes every context forget what it has learnt about its inputs (and about its device's memory)
inline void blind_refresh(Ctx &c)
{
	const u32 epoch = g_env_epoch.load();
	if (c.env_epoch != epoch) {
		c.env_epoch = epoch;
		c.blind_no_room = false;
		for (int k = 0; k < 4; ++k)
			c.blind_skip[k] = c.blind_backoff[k] = 0;
		c.log_skip = c.log_backoff = 0;
		c.boff_forget = true;   // (... and the device-side one of the device-scheduled sorts, SegCtl::boff_*: zeroed by the next attempt)
	}
}
// rsx_reload_env: the context forgets the attempts it lost -- the host's counters above, the device's here (in front of the sample kernel)
inline int blind_forget_device_backoff(Ctx &c)
{
	if (c.boff_forget && c.seg.p) {
		HIP_TRY(hipMemsetAsync(&((SegCtl *)c.seg.p)->boff_skip, 0, 2 * sizeof(u32), c.stream));
		c.boff_forget = false;
	}
	return RSX_OK;
}
// Where the keys-only sorts without a histogram end: 2^30 keys (32-bit offsets in the leaf and tile tables; level-2 slots of more
// than 32 Ki whole keys have no leaf) -- or, 4-byte keys in two-byte slots (rsx_leafc.hpp: slots of up to 40960 values), where the
// mean level-2 slot passes 32 Ki: 2^31 + 65536 keys.  Every offset of such a sort still fits 32 bits: 257 level-1 slots of
// 1.25 x 2^23 keys, 65537 level-2 slots of 40960 values, positions below 2^32.
template <typename KT> size_t blind_keys_end()
{
	if (sizeof(KT) == 4 && dense_cap_max<KT>() >= LEAFC_CAP)
		return (size_t)32769 << 16;
	return (size_t)1 << 30;
}
template <typename KT> bool blind_wanted(Ctx &c, size_t n, size_t payload_bytes = 0, bool rank = false)
{
	if constexpr (sizeof(KT) < 4)
		return false;
	if (env().no_blind || env().no_slack || !hybrid_enabled() || !c.fast || capture_armed() || verify_mode() ||
	    c.small.external || env().no_speculation)
		return false;
	if (n < ((size_t)1 << 22) || n >= (payload_bytes || rank ? (size_t)1 << 30 : blind_keys_end<KT>()))
		return false;
	if (payload_bytes) {
		// 4-byte keys with 4-byte payloads, 16 Mi .. 2^28 pairs.  Round 3: from 96 Mi (one leaf shape, 5120 pairs, whose fixed
		// costs made 16 Mi pairs cost 0.63 ms); with the leaves' three shapes (pairs_blind) -- f32 keys -> ranks / pairs, ms,
		// against one pass per column: 16 Mi 0.271 / 0.273 against 0.271 / 0.299, 32 Mi 0.41 / 0.44 against 0.47 / 0.52, 64 Mi
		// 0.68 / 0.74 against 0.84 / 1.00, 2^27 1.21 / 1.33 (round 3's shape: 1.54 / 1.63)
		// (tools/rank_threshold_probe.py, profiles/r04/rank_threshold_probe.txt); a lower RSX_TWO_LEVEL_MIN_LOG2 (tests) lowers the floor
		// With a wave per leaf for slots of up to 256 / 512 pairs (LeafKCfg<64, 256, 8, 9>, <64, 512, 8, 10>): from 4 Mi pairs -- 8 Mi 0.169 / 0.170 against
		// 0.176 / 0.177 ms, 10 Mi 0.180 / 0.186 against 0.229 / 0.233, 12 Mi 0.194 / 0.201 against 0.240 / 0.248.
		// Round 6: up to 2^29 pairs (slots of 10240 pairs: LeafKCfg<1024, 10240, 4, 13>, fourteen position bits)
		if (sizeof(KT) != 4 || payload_bytes != 4 || n > ((size_t)1 << 29) ||
		    n < std::min((size_t)1 << 22, (size_t)1 << env().two_level_min_log2))   // (4 Mi: 140 against 151 us, 6 Mi 148 against 161)
			return false;
	} else {
		// keys only: without the histogram two levels beat one pass per column earlier than with it.  8-byte keys from 4.5 Mi
		// keys on (a wave per leaf for slots of up to 256 keys, LeafKCfg<64, 256, 8, 9>: 5 Mi 204 against 278 us, 7 Mi 219 against
		// 322, 8 Mi 234 where 128 threads per leaf took 284; five kept columns: 4 Mi 172 against 167, 5 Mi 184 against 207;
		// one level reaches 3-4 Mi keys: tools/u64_small_probe.py), before that from 8 Mi
		// keys on since their leaves come in three shapes (launch_leaves; one shape: from 48 Mi) -- uniform keys 8 Mi 0.287
		// against 0.347 ms, 16 Mi 0.386 against 0.605, 32 Mi 0.58 against 1.18, 64 Mi 0.93 against 2.15; five kept columns: 8 Mi
		// level, 16 Mi 0.337 against 0.404 (tools/u64_threshold_probe.py, profiles/r04/u64_threshold_probe.txt).
		// 4-byte keys, round 4 (their leaves read two-byte slots and are one wave's -- or a row of sixteen lanes' -- work up to
		// 1024 values, rsx_leaf16w_kernel / rsx_leaf16q_kernel): from 7.5 Mi keys (8 Mi 128 against 137 us, tools/size_sweep.py)
		size_t floor_keys = sizeof(KT) == 8 ? (size_t)9 << 19 : (size_t)15 << 19;
		if (env().blind_min_log2)
			floor_keys = (size_t)1 << env().blind_min_log2;
		floor_keys = std::min(floor_keys, (size_t)1 << env().two_level_min_log2);
		if (n < floor_keys)
			return false;
	}
	blind_refresh(c);
	const int kind = blind_kind<KT>(payload_bytes, rank);
	if (c.blind_skip[kind]) {
		--c.blind_skip[kind];
		return false;
	}
	return true;
}

// The device's part: the sample, both passes, the tables and the leaves, enqueued; *enqueued = 0: no room for the slots (or no
// leaf shape for them): nothing was enqueued.  Nothing waits for the host; SegCtl::mode == SEG_MODE_LEAVES (and the pinned
// copy the slack plan writes) says afterwards whether the sort went through.
// ... for a device-scheduled sort (rsx_sort_inplace_async): the same sizes; its back-off lives on the device (SegCtl::boff_skip: nothing is ever read back)
template <typename KT> bool async_blind_ok(Ctx &c, size_t n)
{
	if constexpr (sizeof(KT) < 4)
		return false;
	// (a caller's workspace: only one that was sized for the slots as well, rsx_workspace_bytes_fast)
	if (env().no_blind || env().no_slack || !hybrid_enabled() || !c.fast || capture_armed() || verify_mode() ||
	    (c.small.external && !c.ws_blind) || env().no_speculation)
		return false;
	// (blind_wanted's floors, except that 4-byte keys start at 9 Mi here: at 8 Mi the empty launches of the gated histogram-first
	// kernels behind the attempt make it 153 us against 138 for one pass per column; the blocking sort: 127 against 135-139)
	size_t floor_keys = sizeof(KT) == 8 ? (size_t)1 << 23 : (size_t)9 << 20;
	if (env().blind_min_log2)
		floor_keys = (size_t)1 << env().blind_min_log2;
	floor_keys = std::min(floor_keys, (size_t)1 << env().two_level_min_log2);
	blind_refresh(c);
	return n >= std::max(floor_keys, (size_t)1 << 22) && n < (c.ws_blind ? (size_t)1 << 30 : blind_keys_end<KT>());
}
// ... for key + payload and rank sorts (4-byte keys, 4-byte payloads: blind_wanted's window, without its back-off)
template <typename KT> bool async_pairs_blind_ok(Ctx &c, size_t n, size_t payload_bytes)
{
	if (sizeof(KT) != 4 || payload_bytes != 4)
		return false;
	if (env().no_blind || env().no_slack || !hybrid_enabled() || !c.fast || capture_armed() || verify_mode() || c.small.external ||
	    env().no_speculation)
		return false;
	blind_refresh(c);
	return n >= std::min((size_t)1 << 24, (size_t)1 << env().two_level_min_log2) && n >= ((size_t)1 << 22) && n <= ((size_t)1 << 29);
}

template <typename KT>
int blind_enqueue(Ctx &c, KT *src, KT *aux, size_t n, KdfArgs<KT> ka, int *enqueued)
{
	typedef Sc2Cfg<KT, NoVal> C2;
	*enqueued = 0;
	const u32 mean1 = (u32)(n >> 8), mean2 = (u32)(n >> 16);
	const u32 cap1 = level1_slot_cap<KT>(mean1);
	const u32 cap2 = slot_cap_for(mean2);
	if (cap2 > std::max((u32)LeafShapes<KT>::Big::CAP, dense_cap_max<KT>()))
		return RSX_OK;
	if (c.blind_no_room)
		return RSX_OK;
	// Where the level-1 slots lie.  The attempt only writes after its sample has PROVEN the input unsorted and four columns kept
	// -- from then on the caller's second buffer belongs to the sort whatever route finishes it (radix_sort.hpp:60-62 keeps it
	// untouched only on the early exits) -- so the slots that fit there (n / cap1 of them: 204 of 256) lie there and the library
	// allocates the rest only: 0.25 n keys instead of 1.25 n (2^28 u32 keys: 0.25 GiB + 0.63 GiB of two-byte level-2 slots
	// instead of 1.25 + 1.25).  Needs a slot that holds a tile (a lost attempt's runs go over the slot's own beginning there,
	// rsx_scatter2.hpp); RSX_NO_AUX_SLOTS=1: all slots in scratch memory.
	u32 lo = (aux && !env().no_aux_slots && cap1 >= (u32)C2::TILE) ? (u32)std::min<size_t>(n / cap1, 255) : 0u;
	c.slack_cap = cap2;   // (dense_slots asks for it)
	c.slack_mean = mean2;
	const size_t slot2_bytes = dense_slots<KT>(c) ? 2 : sizeof(KT);
	if (c.slack1.ensure(((size_t)(256 - lo) * cap1 + C2::TILE) * sizeof(KT)) != RSX_OK ||
	    c.slack.ensure(((size_t)65536 * cap2 + C2::TILE) * slot2_bytes) != RSX_OK) {
		// no room for the slots: the ordinary path, now and for this context's later sorts (a multi-GiB hipMalloc that fails
		// is not worth repeating per sort); what was allocated of the pair goes back -- unless this is a device-scheduled sort
		// (AsyncScope): a graph captured earlier may name the old arrays, so nothing is released and nothing is remembered
		(void)hipGetLastError();
		c.slack1_cap = c.slack_cap = 0;
		if (!g_in_async) {
			c.slack1.release();
			c.slack.release();
			c.blind_no_room = true;
		}
		return RSX_OK;
	}
	if (lo) {
		// the level-1 pass reaches both parts with 32-bit element offsets from the lower one (rsx_scatter2.hpp, SegArgs): they
		// must lie within 2^32 elements of each other, the dump area behind the last slot included -- else everything in scratch
		const uintptr_t lo_a = (uintptr_t)aux, hi_a = (uintptr_t)c.slack1.p - (size_t)lo * cap1 * sizeof(KT);
		const uintptr_t span = (std::max(lo_a, hi_a) - std::min(lo_a, hi_a)) / sizeof(KT) + (size_t)257 * cap1 + C2::TILE;
		if (span >= ((uintptr_t)1 << 32)) {
			lo = 0;
			if (c.slack1.ensure(((size_t)256 * cap1 + C2::TILE) * sizeof(KT)) != RSX_OK) {
				(void)hipGetLastError();
				c.slack1_cap = c.slack_cap = 0;
				if (!g_in_async) {
					c.slack1.release();
					c.slack.release();
					c.blind_no_room = true;
				}
				return RSX_OK;
			}
		}
	}
	c.slack1_lo = lo;
	RSX_TRY(seg_layout<KT>(c, n));
	RSX_TRY(c.gscan.ensure(256 * sizeof(u64)));
	const u64 ntiles0 = (n + C2::TILE - 1) / C2::TILE;
	const size_t st_bytes = 256 + (ntiles0 + seg_extra_rows<KT>()) * 256 * 4;
	SegCtl *ctl = (SegCtl *)c.seg.p;
	SegTile *tiles = (SegTile *)((char *)c.seg.p + c.seg_tiles_off);
	LeafSeg *segtab = (LeafSeg *)((char *)c.seg.p + c.seg_segtab_off);
	u32 *btile = (u32 *)((char *)c.seg.p + c.seg_btile_off);
	u64 *off1 = (u64 *)c.gscan.p;
	// (a context in a caller's workspace has neither a pinned control block nor an event: nobody reads a verdict there)
	if (!c.seg_ev && !c.small.external)
		HIP_TRY(hipEventCreateWithFlags(&c.seg_ev, hipEventDisableTiming));
	if (c.host_segctl)
		c.host_segctl->mode = SEG_MODE_NONE;
	c.slack1_cap = cap1;
	c.slack_cap = cap2;
	// 4-byte keys from 64 Mi keys on: the level-1 pass in whole 64-byte atoms (rsx_pass32a_kernel: a workgroup per CU takes a range
	// of tiles and carries what does not fill an atom; a bucket then lies at both ends of its slot)
	const bool atoms = pass16a_wanted<KT>(c);   // (the level-2 pass that writes whole atoms: smaller tiles, two cursors per slot)
	const bool atoms64 = pass64a_narrow_wanted<KT>(c);   // (8-byte keys: the same for the form that writes four-byte slots; the sample decides which form runs)
	typedef Pass32aCfgT<sizeof(KT) == 8 ? 14 : 28> P32;
	const size_t min32 = env().pass32_min_mi ? (size_t)env().pass32_min_mi << 20 : sizeof(KT) == 8 ? (size_t)3 << 23 : (size_t)13 << 22;
	const bool atoms1 = (sizeof(KT) == 4 ? atoms : !env().no_unstable) && !env().no_pass32a && n >= min32 &&
	                    cap1 >= (u32)P32::TILE + 2 * PASS32_BACK;
	// (keys the caller says arrive in order of their top digit, piece by piece: four counters per digit, rsx_pass32.hpp)
	const bool rep4 = (c.hints & 1u) != 0 || (env().probe & 4u) != 0;
	// 8-byte keys in which nothing below the level-1 digit varies above bit 32 (keys below 2^40: BASELINE.json's cfg 3 (ii), (iii)):
	// the level-1 slots can hold low words -- all 256 of them then fit the caller's second buffer -- and the level-2 pass reads four
	// bytes per key.  Both atom passes in both forms are enqueued; the sample decides (SegCtl::narrow == 2).
	c.narrow1 = sizeof(KT) == 8 && atoms1 && atoms64 && lo != 0 && !rep4 && !env().no_narrow1 && (((uintptr_t)aux) & 63) == 0;
	// the sample (workgroup 0: control block, plan) and the zeroing of both passes' status words, one launch
	static_assert(sizeof(SegCtl) <= 256, "the control block is not part of what is zeroed");
	RSX_TRY(blind_forget_device_backoff(c));
	hipLaunchKernelGGL((rsx_blind_precheck_kernel<KT>), dim3(1 + 512), dim3(1024), 0, c.stream, (const KT *)src, (u64)n, ka, ctl,
	                   c.plan(), c.dev_host_plan, (u32x4 *)((char *)c.seg.p + c.seg_status_off), (u64)(2 * st_bytes / 16),
	                   4u,   // (two levels want four kept columns: two for the passes, two or more for the leaves)
	                   // 4-byte keys whose leaves read two-byte slots (rsx_leaf16.hpp): the MSB digits may lie below constant top bits
	                   (u32)(sizeof(KT) == 4 && dense_slots<KT>(c) && !env().no_leaf16 && !env().no_shift ? 1 : 0),
	                   // 8-byte keys in slots rsx_leafk_kernel takes: four-byte slots where the leaves' columns lie in the low word
	                   (u32)(narrow_slots_ok<KT>(cap2) ? (c.narrow1 ? 2 : 1) : 0), 0u,
	                   // a device-scheduled sort keeps its back-off on the device (SegCtl::boff_skip); the blocking sorts keep theirs on the host
	                   (u32)(g_in_async ? 1 : 0), (u32)((env().probe & 4u) ? 1u : c.hints));
	{
		// 4-byte keys: only in front of rsx_pass16a_kernel (a bucket that lies at both ends of its slot is one tile more: that pass's
		// tile table has room for it); from 52 Mi keys, where that pass starts for good -- as first built (the next tile requested
		// ahead, two LDS atomics per key) it was level with the chained pass at 64-80 Mi and 1 % ahead at 96 Mi
		// (profiles/r05/atoms_threshold_probe.txt); as it is now: 0-4 % ahead at 54 .. 95 Mi keys, never behind
		// (tools/ab_sizes.py RSX_PASS32_MIN_MI 96 40 u32 ...), 2-5 % behind in the 1024-value-slot window around 40 Mi.
		// 8-byte keys (atoms of eight keys, 14 Ki-key tiles): in front of the CHAINED level-2 pass, whose status words have a row
		// more per bucket for that (seg_extra_rows); from 24 Mi keys (1.3-2.5 % ahead at 24 .. 44 Mi, level at 20 Mi:
		// tools/ab_sizes.py RSX_PASS32_MIN_MI 48 16 u64 ...) -- tools/ubench/pass32_probe, 2^28 u64 keys: 0.926 ms against 1.01 for
		// the chained pass, 2^27: 0.447 against 0.50.
		if (atoms1) {
			// one base for the stores, the parts' offsets in the slots' places (as launch_seg_pass does for the chained pass)
			u32 off_lo = 0, off_hi = 0;
			KT *kbase = (KT *)c.slack1.p;
			if (lo) {
				const uintptr_t lo_a = (uintptr_t)aux, hi_a = (uintptr_t)c.slack1.p - (size_t)lo * cap1 * sizeof(KT);
				const uintptr_t base_a = std::min(lo_a, hi_a);
				off_lo = (u32)((lo_a - base_a) / sizeof(KT));
				off_hi = (u32)((hi_a - base_a) / sizeof(KT));
				kbase = (KT *)base_a;
			}
			u32 *cur1 = (u32 *)((char *)c.seg.p + c.seg_status_off + st_bytes + 256);
			u32 *ovf = &((SegCtl *)c.seg.p)->overflow;
			const bool plain = ka.fmask == 0 && ka.sflip == 0 && ka.desc == 0;
			ProfScope prof(1, (u64)n * 2 * sizeof(KT), c.stream);
			// (probed and not kept: Pass32aCfgT<12> -- 12 Ki-key tiles, 81 KB of LDS, two workgroups per CU, no prefetch: 0.534-0.543 ms
			// for 2^28 keys where this shape takes 0.470-0.477 on the same box, profiles/r05/pass32a_probe.txt)
			// (the next tile's keys requested while this tile is written out: 1 % ahead at 2^27 keys, 2-5 % BEHIND from 2^28 on -- the
			// level-1 pass of 2^28 keys 0.485 -> 0.457 ms without, three rounds alternating in one process, tools/blind_ab.py;
			// 380 M keys 1.975 -> 1.929 ms, 2^30 5.157 -> 5.130: reads and writes in flight together cost more than the gap between tiles)
			// (8-byte keys: never ahead -- 2^27 keys 0.447 against 0.455 ms, 2^28 0.926 against 0.951)
			// (later, with one LDS atomic per key: never ahead at 54 .. 224 Mi keys either -- 54 Mi 0.340 -> 0.330 ms, 192 Mi 0.983 -> 0.961,
			// level at 80 and 128 Mi, tools/ab_sizes.py RSX_PASS32_PREFETCH 1 0 u32 ...: off unless RSX_PASS32_PREFETCH=1 asks for it)
			const bool prefetch = sizeof(KT) == 4 && env().pass32_prefetch > 0;
#define RSX_LAUNCH_P32R(DIGV, PF, REPV)                                                                                      \
			hipLaunchKernelGGL((rsx_pass32a_kernel<KT, DIGV, PF, P32, REPV>), dim3(256), dim3(P32::BLOCK), 0, c.stream,              \
			                   (const KT *)src, (u64)n, kbase, lo, off_lo, off_hi, cap1, (const SegCtl *)ctl, cur1, ovf, ka)
#define RSX_LAUNCH_P32(DIGV, PF)                                                                                             \
			do {                                                                                                                 \
				if (rep4 && !(PF))                                                                                               \
					RSX_LAUNCH_P32R(DIGV, false, 4);                                                                             \
				else                                                                                                             \
					RSX_LAUNCH_P32R(DIGV, PF, 1);                                                                                \
			} while (0)
			if constexpr (sizeof(KT) == 4) {
				if (plain && prefetch)
					RSX_LAUNCH_P32(DIG_PLAIN, true);
				else if (!plain && prefetch)
					RSX_LAUNCH_P32(DIG_GENERIC, true);
			}
			if (plain && !prefetch)
				RSX_LAUNCH_P32(DIG_PLAIN, false);
			else if (!prefetch)
				RSX_LAUNCH_P32(DIG_GENERIC, false);
#undef RSX_LAUNCH_P32
#undef RSX_LAUNCH_P32R
			if constexpr (sizeof(KT) == 8) {
				if (c.narrow1) {
					// ... and the form that writes low words: slot d = cap1 four-byte places at d x cap1 of the caller's second buffer
					if (plain)
						hipLaunchKernelGGL((rsx_pass32a_kernel<KT, DIG_PLAIN, false, P32, 1, u32>), dim3(256), dim3(P32::BLOCK), 0, c.stream,
						                   (const KT *)src, (u64)n, (u32 *)aux, 256u, 0u, 0u, cap1, (const SegCtl *)ctl, cur1, ovf, ka);
					else
						hipLaunchKernelGGL((rsx_pass32a_kernel<KT, DIG_GENERIC, false, P32, 1, u32>), dim3(256), dim3(P32::BLOCK), 0, c.stream,
						                   (const KT *)src, (u64)n, (u32 *)aux, 256u, 0u, 0u, cap1, (const SegCtl *)ctl, cur1, ovf, ka);
				}
			}
			HIP_TRY(hipGetLastError());
		}
	}
	if (!atoms1)
		RSX_TRY(launch_seg_pass<KT>(c, src, lo ? aux : nullptr, n, ka, -2, 1));
	hipLaunchKernelGGL(rsx_seg_tiles_kernel, dim3(32), dim3(256), 0, c.stream, (const u64 *)c.ghist(), (u64)n, (const Plan *)c.plan(),
	                   atoms ? (u32)Pass16aCfg::TILE : (u32)C2::TILE, tiles, ctl, btile, off1, cap1,
	                   (const u32 *)((char *)c.seg.p + c.seg_status_off + st_bytes + 256), (u32)ntiles0, atoms1 ? PASS32_BACK : 0u,
	                   atoms64 ? (u32)Pass2wCfg<u32>::TILE : 0u, c.narrow1 ? (u32)Pass64aCfgLow::TILE : 0u);
	RSX_TRY(launch_seg_pass<KT>(c, lo ? aux : nullptr, nullptr, n, ka, -2, 2));
	hipLaunchKernelGGL((rsx_seg_slack_plan_kernel<u32>), dim3(256), dim3(256), 0, c.stream,
	                   (const u32 *)((char *)c.seg.p + c.seg_status_off + 256), (const u32 *)btile, (const u64 *)c.ghist(),
	                   (const Plan *)c.plan(), ctl, segtab, cap2, c.dev_host_segctl, (const u64 *)off1,
	                   (atoms ? 2u : atoms64 ? 3u : 1u) | ((env().probe & 1u) << 8));
	HIP_TRY(hipGetLastError());
	if (c.seg_ev)
		HIP_TRY(hipEventRecord(c.seg_ev, c.stream));
	u32 leaf_shape = LeafShapes<KT>::shape_for_slots(cap2);
	if (dense_slots<KT>(c))
		leaf_shape |= 0x100u;   // (the leaves read 2-byte slots: the cut shapes have that variant)
	if (sizeof(KT) == 8 && cap2 <= 5120u)
		leaf_shape |= 0x200u;   // (8-byte keys in slots of up to 5120: rsx_leafk_kernel)
	RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_TWO_LEVEL, leaf_shape, (const u64 *)off1));
	*enqueued = 1;
	return RSX_OK;
}

template <typename KT>
int sort_keys_blind(Ctx &c, KT *src, KT *aux, size_t n, KdfArgs<KT> ka, KT **result, rsx_info *info, int *done)
{
	*done = 0;
	int enqueued = 0;
	const size_t pmark = prof_mark();
	RSX_TRY(blind_enqueue<KT>(c, src, aux, n, ka, &enqueued));
	if (!enqueued)
		return RSX_OK;
	HIP_TRY(hipEventSynchronize(c.seg_ev));
	if (c.host_segctl->mode != SEG_MODE_LEAVES) {
		blind_called_off(c, blind_kind<KT>(0));
		c.slack_cap = 0;
		prof_called_off(pmark, c.stream);
		return RSX_OK;
	}
	c.blind_backoff[blind_kind<KT>(0)] = 0;
	if (sizeof(KT) == 8 && c.host_segctl->narrow) {
		// the sample chose four-byte level-2 slots (SegCtl::narrow): the level-2 pass wrote 4 bytes per key, the leaves read 4
		// (narrow == 2: the level-1 slots hold four bytes per key too)
		const bool n1 = c.host_segctl->narrow == 2u;
		prof_rebook(pmark, c.stream, 2, (u64)n * (4 + sizeof(KT)));
		prof_rebook(pmark, c.stream, 1, (u64)n * ((n1 ? 4 : sizeof(KT)) + 4), 3);   // (the whole-key form of the level-2 pass returned at once)
		if (n1)
			prof_rebook(pmark, c.stream, 1, (u64)n * (sizeof(KT) + 4));            // (what is left of kind 1: the level-1 pass)
	}
	const Plan plan = *c.host_plan;
	info_from_plan(info, plan);
	KT *final = (plan.ncols & 1) ? aux : src;   // radix_sort.hpp:92
	*result = final;
	if (info) {
		info->result_in_aux = final == aux;
		info->hybrid = 5u;
	}
	*done = 1;
	return RSX_OK;
}

// ---- 8-byte keys by (bit length, mantissa) digits: rsx_logroute.hpp ----------------------------------------------------------
// Tried where the sorts without a histogram do not go (their sample said no, or they are backing off): the route's own sample
// says at once whether it is worth the histogram; everything behind it is device-scheduled and the verdict is read once.
template <typename KT> bool log_wanted(Ctx &c, size_t n, const KT *src, const KT *aux)
{
	if constexpr (sizeof(KT) != 8)
		return false;
	if (env().no_log || !hybrid_enabled() || !c.fast || capture_armed() || verify_mode() || c.small.external || env().no_speculation)
		return false;
	// from 24 Mi keys: Zipf-like keys (BASELINE.json's cfg 3 (iv)) against one pass per kept column, one box, tools/log_sizes.py --
	// 16 Mi 0.49 against 0.43 ms (65536 leaf workgroups are a fixed 0.24 ms), 24 Mi 0.55 against 0.61, 32 Mi 0.63 against 0.78,
	// 64 Mi 0.89 against 1.46, 128 Mi 1.43 against 2.65, 256 Mi 2.46 against 5.09 (profiles/r06/log_sizes.txt)
	const size_t floor_keys = env().log_min_log2 ? (size_t)1 << env().log_min_log2 : (size_t)3 << 23;
	// (up to 2^29 + 2^25 keys: a level-1 bucket must fit 256 leaves -- of 5120 values up to 2^28 + 2^24 keys, of 10240 beyond; the
	// heaviest digits of Zipf-like keys hold 1 / 256 of the array -- and larger arrays would pay for the histogram before the plan
	// kernel says no)
	if (n < floor_keys || n > ((size_t)17 << 25))
		return false;
	if (((((uintptr_t)src) & 15) | (((uintptr_t)aux) & 63)) != 0)   // (16-byte loads of the input, 64-byte atoms into aux)
		return false;
	// an attempt that its sample refuses costs a memset, the sample and eight empty launches -- 65 us, 7 % of a sort of 24 Mi keys --,
	// a lost one the histogram and a pass or two: after either the next 1, 3, 7 .. 31 sorts of the context do not ask
	// (tools/log_overhead.py; rsx_reload_env() forgets, as for the sorts without a histogram)
	blind_refresh(c);
	if (c.log_skip) {
		--c.log_skip;
		return false;
	}
	return true;
}

template <typename KT>
int sort_keys_log(Ctx &c, KT *src, KT *aux, size_t n, KdfArgs<KT> ka, KT **result, rsx_info *info, int *done)
{
	*done = 0;
	if constexpr (sizeof(KT) == 8) {
		typedef LogP2Cfg P2;
		const size_t tiles_cap = n / P2::TILE + 257;
		const size_t cur2_off = sizeof(LogCtl), tabs_off = cur2_off + 2 * 65536 * sizeof(u32);
		const size_t tiles_off = (tabs_off + sizeof(LogTabs) + 255) & ~(size_t)255;
		const size_t zero_bytes = tabs_off + offsetof(LogTabs, offs_small);
		const size_t l2_cap = n + n / 8 + (size_t)65536 * 700;   // level-2 slots: values (rsx_log_plan_kernel checks the exact sum)
		if (c.logb.ensure(tiles_off + tiles_cap * sizeof(LogTile)) != RSX_OK ||
		    c.logslots.ensure((l2_cap + P2::TILE + 64) * sizeof(u32)) != RSX_OK) {
			(void)hipGetLastError();
			return RSX_OK;   // (no room: the ordinary path)
		}
		if (!c.host_logctl)
			HIP_TRY(hipHostMalloc((void **)&c.host_logctl, sizeof(LogCtl), hipHostMallocDefault));
		if (!c.log_ev)
			HIP_TRY(hipEventCreateWithFlags(&c.log_ev, hipEventDisableTiming));
		LogCtl *ctl = (LogCtl *)c.logb.p;
		u32 *cur2 = (u32 *)((char *)c.logb.p + cur2_off);
		LogTabs *tabs = (LogTabs *)((char *)c.logb.p + tabs_off);
		LogTile *tiles = (LogTile *)((char *)c.logb.p + tiles_off);
		u32 *slots = (u32 *)c.logslots.p;
		const size_t pmark = prof_mark();
		HIP_TRY(hipMemsetAsync(c.logb.p, 0, zero_bytes, c.stream));
		// the leaves' shape: 5120 values per slot (five workgroups of 256 threads per CU) up to 2^28 + 2^24 keys, 10240 beyond
		const bool big_leaves = n > ((size_t)17 << 24) || env().log_leaf_big;
		hipLaunchKernelGGL((rsx_log_sample_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, (const KT *)src, (u64)n, ka, ctl,
		                   big_leaves ? LOG_LEAF_CAP_BIG : LOG_LEAF_CAP);
		{
			ProfScope prof(0, (u64)n * sizeof(KT), c.stream);
			hipLaunchKernelGGL((rsx_log_hist_kernel<KT>), dim3(512), dim3(1024), 0, c.stream, (const KT *)src, (u64)n, ka, ctl, tabs);
		}
		hipLaunchKernelGGL(rsx_log_plan_kernel, dim3(1), dim3(1024), 0, c.stream, ctl, tabs, tiles, (u64)n, (u32)n, (u32)l2_cap,
		                   (u32)tiles_cap, (u32)P2::TILE, (u32)P2::GRID, (Plan *)nullptr, c.dev_host_plan);
		{
			ProfScope prof(1, 0, c.stream);
			hipLaunchKernelGGL((rsx_log_pass1_kernel<KT>), dim3(256), dim3(LogP1Cfg::BLOCK), 0, c.stream, (const KT *)src, (u64)n, aux,
			                   ctl, tabs, ka);
		}
		{
			ProfScope prof(3, 0, c.stream);
			hipLaunchKernelGGL((rsx_log_pass2_kernel<KT>), dim3(P2::GRID), dim3(P2::BLOCK), 0, c.stream, (const KT *)aux, slots,
			                   (const LogTile *)tiles, ctl, (const LogTabs *)tabs, cur2, (u32)l2_cap, ka);
		}
		{
			ProfScope prof(2, 0, c.stream);
			hipLaunchKernelGGL((rsx_log_fill_kernel<KT>), dim3(2048), dim3(LOG_FILL_BLOCK), 0, c.stream, src, aux, (const LogCtl *)ctl,
			                   (const LogTabs *)tabs, ka);
			if (big_leaves)
				hipLaunchKernelGGL((rsx_log_leaf_kernel<KT, LogLeafCfgBig>), dim3(65536), dim3(LogLeafCfgBig::BLOCK), 0, c.stream, src, aux,
				                   (const u32 *)slots, (const LogCtl *)ctl, (const LogTabs *)tabs, (const u32 *)cur2, ka, 0u, 256u);
			else
				hipLaunchKernelGGL((rsx_log_leaf_kernel<KT, LogLeafCfg>), dim3(65536), dim3(LogLeafCfg::BLOCK), 0, c.stream, src, aux,
				                   (const u32 *)slots, (const LogCtl *)ctl, (const LogTabs *)tabs, (const u32 *)cur2, ka, 0u, 256u);
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(c.host_logctl, ctl, sizeof(LogCtl), hipMemcpyDeviceToHost, c.stream));
		HIP_TRY(hipEventRecord(c.log_ev, c.stream));
		HIP_TRY(hipEventSynchronize(c.log_ev));
		const LogCtl h = *c.host_logctl;
		if (!h.go || h.fail || (!h.ok && !h.sorted)) {
			prof_called_off(pmark, c.stream);   // (not this route's keys, or an attempt that was lost: the ordinary path)
			c.log_backoff = std::min<u32>(2 * c.log_backoff + 1, 31);
			c.log_skip = c.log_backoff;
			return RSX_OK;
		}
		c.log_backoff = 0;
		const Plan plan = *c.host_plan;
		info_from_plan(info, plan);
		*done = 1;
		if (h.sorted) {   // radix_sort.hpp:60-62
			prof_called_off(pmark, c.stream, 1);
			prof_called_off(pmark, c.stream, 2);
			prof_called_off(pmark, c.stream, 3);
			if (info) {
				info->early_exit = 2;
				info->ncols = 0;
			}
			*result = src;
			return RSX_OK;
		}
		const u64 nsmall = h.nsmall, nbig = n - nsmall;
		prof_rebook(pmark, c.stream, 1, (u64)n * sizeof(KT) + nbig * sizeof(KT));
		prof_rebook(pmark, c.stream, 3, nbig * (sizeof(KT) + 4));
		prof_rebook(pmark, c.stream, 2, nbig * (4 + sizeof(KT)) + nsmall * sizeof(KT));
		KT *final = (plan.ncols & 1) ? aux : src;   // radix_sort.hpp:92
		*result = final;
		if (info) {
			info->result_in_aux = final == aux;
			info->hybrid = 6u;
		}
	}
	return RSX_OK;
}

// ---- keys only -------------------------------------------------------------------
template <typename KT>
int sort_keys_device_impl(Ctx &c, KT *src, KT *aux, size_t n, int dtype, int order, void **result, rsx_info *info);

// RSX_VERIFY=2: the sort as it always runs (speculation, leaves, slack slots ...), bracketed by rsx_checksum_kernel on the
// input and on the result: not sorted, or not the same keys -> RSX_EVERIFY.  (RSX_VERIFY=1 re-ranks a tile of every PASS and
// therefore keeps to the pass kernels; this one is blind to where an error came from but covers every route.)
template <typename KT>
int sort_keys_device(Ctx &c, KT *src, KT *aux, size_t n, int dtype, int order, void **result, rsx_info *info)
{
	if (!env().verify_whole)
		return sort_keys_device_impl<KT>(c, src, aux, n, dtype, order, result, info);
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	RSX_TRY(c.vsum.ensure(6 * sizeof(u64)));
	u64 *vs = (u64 *)c.vsum.p;
	HIP_TRY(hipMemsetAsync(vs, 0, 6 * sizeof(u64), c.stream));
	hipLaunchKernelGGL((rsx_checksum_kernel<KT>), dim3(2048), dim3(256), 0, c.stream, (const KT *)src, (u64)n, ka, vs);
	HIP_TRY(hipGetLastError());
	RSX_TRY(sort_keys_device_impl<KT>(c, src, aux, n, dtype, order, result, info));
	hipLaunchKernelGGL((rsx_checksum_kernel<KT>), dim3(2048), dim3(256), 0, c.stream, (const KT *)*result, (u64)n, ka, vs + 3);
	HIP_TRY(hipGetLastError());
	u64 h[6];
	HIP_TRY(hipMemcpyAsync(h, vs, sizeof h, hipMemcpyDeviceToHost, c.stream));
	HIP_TRY(hipStreamSynchronize(c.stream));
	if (env().verify_inject)   // (test hook: the failure report end to end)
		h[5] ^= 1;
	if (h[3] != 0 || h[1] != h[4] || h[2] != h[5])
		return fail(RSX_EVERIFY, "RSX_VERIFY=2: the result of a sort of %zu keys (route %u) is %s: %llu descents, key sum %s, key mix %s",
		            n, info ? info->hybrid : 0u, h[3] ? "not sorted" : "not a permutation of the input", (unsigned long long)h[3],
		            h[1] == h[4] ? "kept" : "changed", h[2] == h[5] ? "kept" : "changed");
	return RSX_OK;
}

template <typename KT>
int sort_keys_device_impl(Ctx &c, KT *src, KT *aux, size_t n, int dtype, int order, void **result, rsx_info *info)
{
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (c.fast && n * sizeof(KT) <= SMALL_SORT_BYTES && !env().no_small_sort && !capture_armed()) {
		// the whole sort in one workgroup and one launch (rsx_small.hpp)
		ProfScope prof(1, (u64)n * 2 * sizeof(KT), c.stream);
		hipLaunchKernelGGL((rsx_small_sort_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, src, aux, (u32)n, ka, c.dev_host_plan);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(c.stream));
		const Plan p = *c.host_plan;
		info_from_plan(info, p);
		if (info && p.sorted)
			info->early_exit = 2;
		*result = (p.ncols & 1) ? aux : src;     // radix_sort.hpp:92 (sorted input: no column, src)
		if (info)
			info->result_in_aux = *result == aux;
		return RSX_OK;
	}
	Plan plan;
	if constexpr (sizeof(KT) == 1) {
		// 1-byte keys: at most one column, so the sorted array is the histogram written out (rsx_fill_runs_kernel) -- a read
		// and a write of the keys instead of a read, a read and a scatter
		if (!env().no_fill_runs && (((uintptr_t)aux) & 15) == 0) {
			RSX_TRY(plan_phase<KT>(c, src, n, ka, nullptr, 0));
			const unsigned blocks = (unsigned)std::min<u64>((n / 16 + 255) / 256 + 1, 8192);
			hipLaunchKernelGGL((rsx_fill_runs_kernel<KT>), dim3(blocks), dim3(256), 0, c.stream, aux, (u64)n, (const u64 *)c.ghist(),
			                   (const KT *)src, ka, (const Plan *)c.plan());
			HIP_TRY(hipGetLastError());
			RSX_TRY(plan_wait(c, &plan));
			info_from_plan(info, plan);
			RSX_TRY(capture_hist(c, n, sizeof(KT)));
			if (plan.sorted) {                   // radix_sort.hpp:60-62
				if (info) {
					info->early_exit = 2;
					info->ncols = 0;
				}
				*result = src;
				return RSX_OK;
			}
			*result = aux;                       // one pass: radix_sort.hpp:92
			if (info)
				info->result_in_aux = 1;
			return RSX_OK;
		}
	}
	if constexpr (sizeof(KT) >= 4) {
		// large arrays: two MSB passes and leaves without the histogram, where a sample of the keys allows it (rsx_hybrid.hpp)
		if (blind_wanted<KT>(c, n)) {
			int done = 0;
			KT *res = nullptr;
			RSX_TRY(sort_keys_blind<KT>(c, src, aux, n, ka, &res, info, &done));
			if (done) {
				*result = res;
				return RSX_OK;
			}
		}
	}
	if constexpr (sizeof(KT) == 8) {
		// 8-byte keys the byte columns do not spread (heavy-tailed magnitudes): digits of (bit length, mantissa), rsx_logroute.hpp
		if (log_wanted<KT>(c, n, src, aux)) {
			int done = 0;
			KT *res = nullptr;
			RSX_TRY(sort_keys_log<KT>(c, src, aux, n, ka, &res, info, &done));
			if (done) {
				*result = res;
				return RSX_OK;
			}
		}
	}
	// The first pass is enqueued before the host knows the plan (it reads the device's copy and does nothing on
	// sorted input): the host's wait for the plan, 20-25 us of idle GPU otherwise, hides behind it.
	const bool spec = c.fast && !env().no_speculation && !verify_mode();
	// with the fast kernel every pass has its own region of status words, all zeroed together with the histogram
	const size_t status_total = c.fast ? status_bytes<KT, NoVal>(n) * sizeof(KT) : 0;
	if constexpr (sizeof(KT) == 2) {
		// 2-byte keys, large arrays: one 16-bit digit.  The sorted array is written from the joint histogram of the two bytes
		// (rsx_joint16_kernel ... rsx_fill16_kernel) into `src` -- where two passes end (radix_sort.hpp:92) --, from one
		// byte's histogram into `aux` if only one column is kept; the device-side plan decides, no kernel scatters.
		// (below 2^32 keys: the joint table counts in 32 bits, and one 16-bit value may occur n times; larger arrays take the
		// two scatter passes, whose status words are 64-bit from 2^30 keys on -- counter width by n, radix_sort.hpp:102-114)
		if (!env().no_fill_runs && n >= ((size_t)1 << 20) && n < ((size_t)1 << 32) && ((((uintptr_t)aux) | ((uintptr_t)src)) & 15) == 0) {
			RSX_TRY(c.joint.ensure(65536 * sizeof(u32) + 65537 * sizeof(u64) + 8));
			u32 *jt = (u32 *)c.joint.p;
			u64 *offs = (u64 *)((char *)c.joint.p + 65536 * sizeof(u32));
			HIP_TRY(hipMemsetAsync(jt, 0, 65536 * sizeof(u32), c.stream));
			RSX_TRY(plan_phase<KT>(c, src, n, ka, nullptr, 0));
			hipLaunchKernelGGL(rsx_joint16_kernel, dim3(512), dim3(1024), 0, c.stream, (const uint16_t *)src, (u64)n, ka, jt,
			                   (const Plan *)c.plan());
			hipLaunchKernelGGL(rsx_joint16_scan_kernel, dim3(1), dim3(1024), 0, c.stream, (const u32 *)jt, offs, (u64)n,
			                   (const Plan *)c.plan());
			const unsigned blocks = (unsigned)std::min<u64>((n / 8 + 255) / 256 + 1, 8192);
			hipLaunchKernelGGL(rsx_fill16_kernel, dim3(blocks), dim3(256), 0, c.stream, (uint16_t *)src, (u64)n, (const u64 *)offs, ka,
			                   (const Plan *)c.plan());
			hipLaunchKernelGGL((rsx_fill_runs_kernel<KT>), dim3(blocks), dim3(256), 0, c.stream, aux, (u64)n, (const u64 *)c.ghist(),
			                   (const KT *)src, ka, (const Plan *)c.plan());
			HIP_TRY(hipGetLastError());
			RSX_TRY(plan_wait(c, &plan));
			info_from_plan(info, plan);
			RSX_TRY(capture_hist(c, n, sizeof(KT)));
			if (plan.sorted) {                   // radix_sort.hpp:60-62
				if (info) {
					info->early_exit = 2;
					info->ncols = 0;
				}
				*result = src;
				return RSX_OK;
			}
			*result = plan.ncols == 1 ? aux : src;
			if (info)
				info->result_in_aux = plan.ncols == 1;
			return RSX_OK;
		}
	}
	// One kept column (keys that differ in one byte only): the sorted array is written from the histogram instead of
	// scattered (rsx_fill_runs_kernel).  With a speculative first pass both kernels are enqueued and the device-side plan
	// decides which of them works, which costs an empty launch (3 us) in the usual case: only from 16 Mi keys on, where that is 1 %.
	const bool fill_one = sizeof(KT) > 1 && !env().no_fill_runs && (((uintptr_t)aux) & 15) == 0 && (!spec || n >= ((size_t)1 << 24));
	auto launch_fill = [&]() {
		const unsigned blocks = (unsigned)std::min<u64>((n * sizeof(KT) / 16 + 255) / 256 + 1, 8192);
		hipLaunchKernelGGL((rsx_fill_runs_kernel<KT>), dim3(blocks), dim3(256), 0, c.stream, aux, (u64)n, (const u64 *)c.ghist(),
		                   (const KT *)src, ka, (const Plan *)c.plan());
		return hipGetLastError();
	};
	u32 spec_leaves = 0;
	bool self_planned = false;
	const size_t pmark = prof_mark();
	if (spec) {
		// (the device may choose one MSB pass and leaves, rsx_hybrid.hpp: pass 0 then goes by the highest kept column)
		const HybCaps caps = capture_armed() ? HybCaps{0, 0, 0, 0} : hybrid_caps<KT>(n);
		// Mid-size arrays: pass 0 derives the plan itself (SCATTER_SELF_PLAN, rsx_scatter2.hpp) -- no plan launch.  (Not
		// with a caller's histogram: its counts are read back from the scanned offsets a plan kernel leaves.)
		self_planned = sizeof(KT) >= 4 && caps.cap1 != 0 && n <= ((size_t)1 << 23) && !fill_one && !env().no_self_plan;
		RSX_TRY(plan_phase<KT>(c, src, n, ka, nullptr, status_total, caps, true, &self_planned));
		u32 flags0 = fill_one ? (u32)SCATTER_ONE_COL_FILLED : 0u;
		if (self_planned) {
			flags0 |= SCATTER_SELF_PLAN;
			c.pass_sp = SelfPlanArgs{(const u32 *)c.unsorted(), c.plan(), c.dev_host_plan, (u64 *)c.gscan.p, caps};
		}
		const int rc0 = scatter_pass<KT, NoVal>(c, src, aux, nullptr, nullptr, n, 0, c.ghist(), ka, flags0, c.plan(), 0);
		c.pass_sp = SelfPlanArgs{nullptr, nullptr, nullptr, nullptr, HybCaps{0, 0, 0, 0}};
		RSX_TRY(rc0);
		if (self_planned) {   // the plan is pass 0's workgroup 0's now
			if (!c.plan_ev)
				HIP_TRY(hipEventCreateWithFlags(&c.plan_ev, hipEventDisableTiming));
			HIP_TRY(hipEventRecord(c.plan_ev, c.stream));
		}
		if (fill_one)
			HIP_TRY(launch_fill());
		if constexpr (sizeof(KT) >= 4) {
			// one MSB pass and leaves, if the device-side plan says so: enqueued now, so that nothing waits for the host
			// (the large shape only where even spread keys would come near the small one's capacity; else after the wait)
			if (caps.cap1 && n <= (size_t)256 * caps.cap1) {
				spec_leaves = n / 256 > (size_t)LeafShapes<KT>::Small::CAP / 2 ? (LeafShapes<KT>::HAS_MEDIUM ? 7u : 3u) : 1u;
				RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_ONE_LEVEL, spec_leaves, self_planned ? (const u64 *)c.gscan.p : nullptr));
			}
		}
		RSX_TRY(plan_wait(c, &plan));
	} else {
		RSX_TRY(plan_phase<KT>(c, src, n, ka, &plan, status_total));
	}
	info_from_plan(info, plan);
	RSX_TRY(capture_hist(c, n, sizeof(KT)));
	// (the profile books what the device chose: leaves enqueued for a plan that did not come, a pass 0 that found the input sorted)
	if (spec_leaves && (plan.sorted || plan.hyb != HYB_ONE_LEVEL))
		prof_called_off(pmark, c.stream, 2);
	if (spec && plan.sorted)
		prof_called_off(pmark, c.stream, 1);
	if (plan.sorted) {                       // radix_sort.hpp:60-62
		if (info) {
			info->early_exit = 2;
			info->ncols = 0;
		}
		*result = src;
		return RSX_OK;
	}
	if (fill_one && plan.ncols == 1) {
		if (!spec)
			HIP_TRY(launch_fill());
		*result = aux;                       // one pass: radix_sort.hpp:92
		if (info)
			info->result_in_aux = 1;
		return RSX_OK;
	}
	if constexpr (sizeof(KT) >= 4) {
		if (plan.hyb == HYB_ONE_LEVEL || plan.hyb == HYB_TWO_LEVEL) {
			// pass 0 went by the highest kept column; the leaves put the result where an LSB-first sort of plan.ncols passes ends
			KT *final = (plan.ncols & 1) ? aux : src;
			u32 how = 1;
			if (plan.hyb == HYB_TWO_LEVEL) {
				RSX_TRY(sort_keys_two_level<KT>(c, src, aux, n, ka, plan, &final, &how));
			} else {
				// one level: the leaves are on their way, unless they need a shape that was not enqueued
				const u32 need = LeafShapes<KT>::shape_for(plan.max1);
				if (!(spec_leaves & need))
					RSX_TRY(launch_leaves<KT>(c, src, aux, n, ka, HYB_ONE_LEVEL, need, self_planned ? (const u64 *)c.gscan.p : nullptr));
			}
			*result = final;                     // radix_sort.hpp:92
			if (info) {
				info->result_in_aux = final == aux;
				info->hybrid = how;
			}
			return RSX_OK;
		}
	}
	if (self_planned && plan.ncols > 1) {
		// one pass per kept column after a self-planned pass 0: the other columns' scans (and the hot digits) are made now
		hipLaunchKernelGGL((rsx_plan_all_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, (const KT *)src, (u64)n, c.ghist(), ka, c.kept(),
		                   c.hotd(), (const u32 *)c.unsorted(), c.plan(), c.dev_host_plan, HybCaps{0, 0, 0, 0});
		HIP_TRY(hipGetLastError());
	}
	KT *cur = src, *oth = aux;
	if (spec)
		std::swap(cur, oth);                 // pass 0 is on its way
	for (u32 i = spec ? 1 : 0; i < plan.ncols; ++i) {   // radix_sort.hpp:83-90
		const u32 col = plan.cols[i];
		RSX_TRY((scatter_pass<KT, NoVal>(c, cur, oth, nullptr, nullptr, n, 8 * col, c.ghist() + 256 * col, ka,
		                                 hot_flags(plan.hot, col), nullptr, c.fast ? (int)i : -1)));
		std::swap(cur, oth);
	}
	*result = cur;                           // radix_sort.hpp:92
	if (info)
		info->result_in_aux = cur == aux;
	return RSX_OK;
}

// ---- keys only, no host synchronisation: every pass is device-scheduled, the result always ends in `buf` ------------
template <typename KT>
int sort_keys_inplace_async(Ctx &c, KT *buf, KT *scratch, size_t n, int dtype, int order)
{
	if (!c.fast)
		return fail(RSX_EHIP, "rsx_sort_inplace_async needs the fast scatter kernel (the device self-check failed on this device)");
	c.async_tried_blind = false;   // (rsx_async_route reports THIS call: set again below if an attempt is enqueued)
	c.async_small = false;
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (n * sizeof(KT) <= SMALL_SORT_BYTES) {
		hipLaunchKernelGGL((rsx_small_sort_kernel<KT>), dim3(1), dim3(1024), 0, c.stream, buf, scratch, (u32)n, ka, c.dev_host_plan, true);
		HIP_TRY(hipGetLastError());
		c.async_small = true;
		return RSX_OK;
	}
	const size_t status_total = status_bytes<KT, NoVal>(n) * sizeof(KT);
	// The routes of the blocking sort (rsx_hybrid.hpp), chosen on the device with nobody to read a verdict back:
	//  * arrays the sort without a histogram is for: the whole attempt is enqueued first -- sample, two MSB passes
	//    into slots, leaves -- and what follows (histogram, plan, one pass per kept column) looks at the attempt's verdict,
	//    SegCtl::mode, and does nothing if the keys are sorted by then; an attempt that is called off has only read `buf`.
	//    (The host never learns how an attempt went: a context's back-off after a LOST attempt -- one that passed the sample and
	//    then overflowed a slot -- is kept on the device, SegCtl::boff_skip.  An attempt the sample turns away costs a sample kernel
	//    and a few empty launches, one that goes through the empty launches of the histogram-first kernels.)
	//  * mid-size arrays: one MSB pass and leaves where the device-side plan says so (Plan::hyb) -- pass 0 then goes by the
	//    highest kept column, the leaf launches behind it do nothing otherwise, and the passes 1 .. do nothing if they do.
	// A caller-owned workspace (rsx_sort_inplace_async_ws) makes the attempt if it was sized for the slots (rsx_workspace_bytes_fast).
	HybCaps caps{0, 0, 0, 0};
	int blind = 0;
	ProfAsyncVerdict pverdict(c.stream);   // (rsx_profile books what the device chose: the attempt's launches or the ones behind it)
	if constexpr (sizeof(KT) >= 4) {
		if (hybrid_enabled() && !verify_mode() && !env().no_speculation) {
			caps = hybrid_caps<KT>(n);
			caps.cap2 = caps.min_cols2 = 0;
			if (async_blind_ok<KT>(c, n))
				RSX_TRY(blind_enqueue<KT>(c, buf, scratch, n, ka, &blind));
		}
	}
	if (blind)
		pverdict.attempt_enqueued((const SegCtl *)c.seg.p);
	c.pass_gate = blind ? (const SegCtl *)c.seg.p : nullptr;
	c.async_tried_blind = blind != 0;
	int rc = plan_phase<KT>(c, buf, n, ka, nullptr, status_total, caps);
	for (u32 i = 0; i < sizeof(KT) && rc == RSX_OK; ++i)   // pass i = the i-th kept column, if there is one (radix_sort.hpp:83-90)
		rc = scatter_pass<KT, NoVal>(c, buf, scratch, nullptr, nullptr, n, 0, c.ghist(), ka, 0, c.plan(), (int)i, i);
	c.pass_gate = nullptr;
	RSX_TRY(rc);
	if constexpr (sizeof(KT) >= 4) {
		if (caps.cap1 && n <= (size_t)256 * caps.cap1) {
			// (every shape: which one the largest bucket needs is only known on the device, and each does nothing unless the
			// plan's largest bucket is its size -- keys with 64 values in their top byte fill buckets four times the mean)
			const u32 shapes = LeafShapes<KT>::HAS_MEDIUM ? 7u : 3u;
			RSX_TRY(launch_leaves<KT>(c, buf, scratch, n, ka, HYB_ONE_LEVEL, shapes));
		}
	}
	pverdict.gated_enqueued();
	// an odd number of kept columns leaves the result in `scratch` (radix_sort.hpp:92): bring it home
	hipLaunchKernelGGL(rsx_copy_if_odd_kernel, dim3(2048), dim3(256), 0, c.stream, (unsigned char *)buf, (const unsigned char *)scratch,
	                   (u64)n * sizeof(KT), (const Plan *)c.plan());
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// ---- key + payload, no host synchronisation: as sort_keys_inplace_async, the result always in (k, v) -----------------------
template <typename KT, typename VT>
int pairs_blind_enqueue(Ctx &c, const KT *kin, const VT *vin, KT *kfinal, VT *vfinal, size_t n, KdfArgs<KT> ka, int *enqueued,
                        KT *kspare = nullptr, VT *vspare = nullptr);

template <typename KT, typename VT>
int sort_pairs_inplace_async(Ctx &c, KT *k, KT *ks, VT *v, VT *vs, size_t n, int dtype, int order)
{
	if (!c.fast)
		return fail(RSX_EHIP, "rsx_sort_pairs_inplace_async needs the fast scatter kernel (the device self-check failed on this device)");
	c.async_tried_blind = false;   // (rsx_async_route reports THIS call: set again below if an attempt is enqueued)
	c.async_small = false;
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (n * 2 * (sizeof(KT) + sizeof(VT)) <= SMALL_PAIR_BYTES) {
		hipLaunchKernelGGL((rsx_small_pairs_kernel<KT, VT, false>), dim3(1), dim3(1024), 0, c.stream, (const KT *)k, ks, v, vs, (u32)n,
		                   ka, c.dev_host_plan, true);
		HIP_TRY(hipGetLastError());
		c.async_small = true;
		return RSX_OK;
	}
	const size_t status_total = status_bytes<KT, VT>(n) * sizeof(KT);
	// as sort_keys_inplace_async: the attempt without a histogram first (4-byte keys and payloads, 16 Mi .. 2^28 pairs: two MSB
	// passes into slots and the pairs' leaves, which write (k, v) -- an attempt that is called off has only read them), the
	// histogram-first kernels behind it gated on its verdict
	int blind = 0;
	ProfAsyncVerdict pverdict(c.stream);
	if constexpr (sizeof(KT) == 4 && sizeof(VT) == 4) {
		if (async_pairs_blind_ok<KT>(c, n, sizeof(VT)))
			RSX_TRY((pairs_blind_enqueue<KT, VT>(c, k, v, k, v, n, ka, &blind, ks, vs)));
	}
	if (blind)
		pverdict.attempt_enqueued((const SegCtl *)c.seg.p);
	c.pass_gate = blind ? (const SegCtl *)c.seg.p : nullptr;
	c.async_tried_blind = blind != 0;
	int rc = plan_phase<KT>(c, k, n, ka, nullptr, status_total);
	for (u32 i = 0; i < sizeof(KT) && rc == RSX_OK; ++i)
		rc = scatter_pass<KT, VT>(c, k, ks, v, vs, n, 0, c.ghist(), ka, 0, c.plan(), (int)i, i);
	c.pass_gate = nullptr;
	RSX_TRY(rc);
	pverdict.gated_enqueued();
	hipLaunchKernelGGL(rsx_copy_if_odd_kernel, dim3(2048), dim3(256), 0, c.stream, (unsigned char *)k, (const unsigned char *)ks,
	                   (u64)n * sizeof(KT), (const Plan *)c.plan());
	hipLaunchKernelGGL(rsx_copy_if_odd_kernel, dim3(2048), dim3(256), 0, c.stream, (unsigned char *)v, (const unsigned char *)vs,
	                   (u64)n * sizeof(VT), (const Plan *)c.plan());
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// ---- two MSB passes and leaves for key + payload sorts and rank sorts (4-byte keys, 4-byte payloads; rsx_leaf_pairs_kernel) ----
template <typename KT> HybCaps hybrid_caps_pairs(size_t n, size_t val_bytes_)
{
	HybCaps caps{0, 0, 0, 0};
	if (sizeof(KT) != 4 || val_bytes_ != 4 || !hybrid_enabled())
		return caps;
	// one level where every bucket of the highest kept column fits the pairs' leaf (5120 pairs: up to about a million pairs)
	caps.cap1 = 5120;
	caps.min_cols1 = 3;
	// two levels: the slack route only, and only where a slot fits the pairs' leaf shape: 2^27 .. 2^28 pairs (cfg 4)
	if (!env().no_slack && n >= ((size_t)1 << env().two_level_min_log2) && n <= ((size_t)1 << 28)) {
		caps.cap2 = (u32)LeafShapes<KT>::Small::CAP;
		caps.min_cols2 = 4;
	}
	return caps;
}

// Pass 1 (by the highest kept column) has written (k1, v1).  The second pass goes into slots, the leaves write the payloads
// (and the keys, if kfinal) to (kfinal, vfinal).  *ok = false: a slot overflowed -- nothing the caller owns was written, it
// sorts with one pass per column.
template <typename KT, typename VT>
int pairs_two_level(Ctx &c, const KT *k1, const VT *v1, KT *kfinal, VT *vfinal, size_t n, KdfArgs<KT> ka, bool *ok)
{
	typedef Sc2Cfg<KT, VT> C2;
	typedef LeafCfg<u32, 4, 20, 3> L;   // 5120 pairs: the slack slot of 2^28 pairs; three workgroups per CU
	*ok = false;
	const u64 rows = (n + C2::TILE - 1) / C2::TILE + 256;
	const size_t st_bytes = 256 + rows * 256 * 4;
	const size_t hist_bytes = (size_t)256 * (sizeof(KT) - 1) * 256 * sizeof(u32);
	c.seg_hist_off = 256;
	c.seg_status_off = c.seg_hist_off + hist_bytes;
	c.seg_segtab_off = c.seg_status_off + (sizeof(KT) - 1) * st_bytes;
	c.seg_tiles_off = c.seg_segtab_off + 65536 * sizeof(LeafSeg);
	c.seg_btile_off = c.seg_tiles_off + rows * sizeof(SegTile);
	{
		const void *before = c.seg.p;
		RSX_TRY(c.seg.ensure(c.seg_btile_off + 257 * sizeof(u32)));
		if (c.seg.p != before)
			HIP_TRY(hipMemsetAsync(c.seg.p, 0, 256, c.stream));
	}
	const u32 mean = (u32)(n >> 16);
	const u32 cap = slot_cap_for(mean);
	if (cap > (u32)L::CAP)
		return RSX_OK;
	if (c.slack.ensure(((size_t)65536 * cap + C2::TILE) * sizeof(KT)) != RSX_OK ||
	    c.slack_v.ensure(((size_t)65536 * cap + C2::TILE) * sizeof(VT)) != RSX_OK) {
		(void)hipGetLastError();
		return RSX_OK;   // (no room for the slots: one pass per column)
	}
	SegCtl *ctl = (SegCtl *)c.seg.p;
	SegTile *tiles = (SegTile *)((char *)c.seg.p + c.seg_tiles_off);
	LeafSeg *segtab = (LeafSeg *)((char *)c.seg.p + c.seg_segtab_off);
	u32 *btile = (u32 *)((char *)c.seg.p + c.seg_btile_off);
	if (!c.seg_ev)
		HIP_TRY(hipEventCreateWithFlags(&c.seg_ev, hipEventDisableTiming));
	HIP_TRY(hipMemsetAsync(c.seg.p, 0, c.seg_status_off + st_bytes, c.stream));
	hipLaunchKernelGGL(rsx_seg_tiles_kernel, dim3(32), dim3(256), 0, c.stream, (const u64 *)c.ghist(), (u64)n, (const Plan *)c.plan(),
	                   (u32)C2::TILE, tiles, ctl, btile);
	char *base = (char *)c.seg.p + c.seg_status_off;
	SegArgs sa{};
	sa.ctl = ctl;
	sa.hist = (const u32 *)((char *)c.seg.p + c.seg_hist_off);
	sa.tiles = tiles;
	sa.slots = (u32)sizeof(KT) - 1;
	sa.slack_cap = cap;
	sa.overflow = &ctl->overflow;
	{
		ProfScope prof(1, (u64)n * 2 * (sizeof(KT) + sizeof(VT)), c.stream);
		hipLaunchKernelGGL((rsx_scatter2_kernel<KT, VT, u32, C2, false, DIG_GENERIC, false, KT, true>), dim3((unsigned)rows),
		                   dim3(C2::BLOCK), 0, c.stream, k1, (KT *)c.slack.p, v1, (VT *)c.slack_v.p, (u64)n, 0u, (const u64 *)c.ghist(), 1u,
		                   (u32 *)(base + 256), (u32 *)base, ka, (u32)SCATTER_SEG_SLACK, (u64 *)nullptr, (const Plan *)c.plan(), 0u, 0u,
		                   (const u32 *)nullptr, sa);
	}
	hipLaunchKernelGGL((rsx_seg_slack_plan_kernel<u32>), dim3(256), dim3(256), 0, c.stream, (const u32 *)(base + 256), (const u32 *)btile,
	                   (const u64 *)c.ghist(), (const Plan *)c.plan(), ctl, segtab, cap, c.dev_host_segctl);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c.seg_ev, c.stream));
	{
		ProfScope prof(2, (u64)n * (sizeof(KT) + 2 * sizeof(VT) + (kfinal ? sizeof(KT) : 0)), c.stream);
		hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, L>), dim3(env().leaf_grid), dim3(L::BLOCK), 0, c.stream, (const KT *)c.slack.p,
		                   (const VT *)c.slack_v.p, cap, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab,
		                   (const SegCtl *)ctl, ka);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventSynchronize(c.seg_ev));
	*ok = c.host_segctl->mode == SEG_MODE_LEAVES;
	return RSX_OK;
}

// One MSB pass has written (k1, v1); the 256 buckets' pairs sorted by the remaining columns into (kfinal, vfinal).
template <typename KT, typename VT>
int pairs_one_level(Ctx &c, const KT *k1, const VT *v1, KT *kfinal, VT *vfinal, size_t n, KdfArgs<KT> ka)
{
	typedef LeafCfg<u32, 4, 20, 3> L;
	ProfScope prof(2, (u64)n * (sizeof(KT) + 2 * sizeof(VT) + (kfinal ? sizeof(KT) : 0)), c.stream);
	hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, L>), dim3(256), dim3(L::BLOCK), 0, c.stream, k1, v1, 0u, kfinal, vfinal,
	                   (const Plan *)c.plan(), (const LeafSeg *)nullptr, (const SegCtl *)nullptr, ka, (u32)HYB_ONE_LEVEL,
	                   (const u64 *)c.ghist(), (u64)n);
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

// Rank sorts and key + payload sorts without the histogram (sort_keys_blind's scheme with the (key, payload) pass kernel and
// the pairs' leaves): both MSB passes into slots -- the first reads the caller's (kin, vin), or makes the indices (vin ==
// nullptr) --, the leaves write to (kfinal, vfinal).  *done = 0: called off, nothing the caller owns has been written.
template <typename KT, typename VT>
int pairs_blind_enqueue(Ctx &c, const KT *kin, const VT *vin, KT *kfinal, VT *vfinal, size_t n, KdfArgs<KT> ka, int *enqueued,
                        KT *kspare, VT *vspare)
{
	typedef Sc2Cfg<KT, VT> C2;
	typedef LeafCfg<u32, 4, 20, 3> L;
	// The level-2 slots hold the low two bytes of what the level-2 pass reads (derived keys, or the packed keys of SegCtl::compact):
	// the two MSB passes have decided every bit above them, and no leaf ever looked at more of a key (rsx_leafp_kernel, K16)
	typedef unsigned short K2;
	static_assert(sizeof(KT) == 4, "two MSB digits above two bytes");
	*enqueued = 0;
	const u32 mean1 = (u32)(n >> 8), mean2 = (u32)(n >> 16);
	const u32 cap1 = slot_cap_for(mean1), cap2 = slot_cap_for(mean2);
	typedef LeafCfg<u32, 16, 12, 1> LB;          // ... and up to 12288 (slots of 10240 pairs: 2^28 .. 2^29 pairs), one workgroup per CU
	typedef LeafKCfg<1024, 10240, 4, 13> P10;   // the compound leaves' shape for those slots
	if (cap2 > (u32)P10::CAP)
		return RSX_OK;
	const bool big = cap2 > (u32)L::CAP || env().pairs_leaf_big;
	if (c.blind_no_room)
		return RSX_OK;
	// Where the level-1 slots lie (as blind_enqueue: nothing is written before the sample has proven the input unsorted and every
	// column kept, after which the caller's spare buffers belong to the sort whatever route finishes it).  `kspare` / `vspare`: n
	// elements each that the attempt may use -- the second key and payload buffers of a key + payload sort; of a rank sort the
	// two halves of its index buffer (the first for the indices until the leaves write it, the second, through which the
	// reference's passes ping-pong, for the 4-byte keys).  The slots that fit (n / cap1 of them: 204 of 256) lie there, the
	// others in scratch; keys and payloads split at the same slot, either spare buffer may be missing.
	u32 lo = (!env().no_aux_slots && cap1 >= (u32)C2::TILE && (kspare || vspare)) ? (u32)std::min<size_t>(n / cap1, 255) : 0u;
	auto part_span = [&](const void *spare, const void *scratch, size_t esz) {   // both parts within 2^32 elements of the lower one?
		const uintptr_t lo_a = (uintptr_t)spare, hi_a = (uintptr_t)scratch - (size_t)lo * cap1 * esz;
		return (std::max(lo_a, hi_a) - std::min(lo_a, hi_a)) / esz + (size_t)257 * cap1 + C2::TILE;
	};
	for (int attempt = 0; attempt < 2; ++attempt) {
		const u32 klo = kspare ? lo : 0u, vlo = vspare ? lo : 0u;
		if (c.slack1.ensure(((size_t)(256 - klo) * cap1 + C2::TILE) * sizeof(KT)) != RSX_OK ||
		    c.slack1_v.ensure(((size_t)(256 - vlo) * cap1 + C2::TILE) * sizeof(VT)) != RSX_OK)
			break;   // (the test below sees it)
		if (lo && ((klo && part_span(kspare, c.slack1.p, sizeof(KT)) >= ((uintptr_t)1 << 32)) ||
		           (vlo && part_span(vspare, c.slack1_v.p, sizeof(VT)) >= ((uintptr_t)1 << 32)))) {
			lo = 0;   // too far apart for 32-bit element offsets: all slots in scratch
			continue;
		}
		break;
	}
	const u32 klo = kspare ? lo : 0u, vlo = vspare ? lo : 0u;
	if (c.slack1.ensure(((size_t)(256 - klo) * cap1 + C2::TILE) * sizeof(KT)) != RSX_OK ||
	    c.slack1_v.ensure(((size_t)(256 - vlo) * cap1 + C2::TILE) * sizeof(VT)) != RSX_OK ||
	    c.slack.ensure(((size_t)65536 * cap2 + C2::TILE) * sizeof(K2)) != RSX_OK ||
	    c.slack_v.ensure(((size_t)65536 * cap2 + C2::TILE) * sizeof(VT)) != RSX_OK) {
		(void)hipGetLastError();   // (no room: as blind_enqueue -- what was allocated goes back, nobody asks again)
		c.slack1_cap = c.slack_cap = 0;
		if (!g_in_async) {
			c.slack1.release();
			c.slack1_v.release();
			c.slack.release();
			c.slack_v.release();
			c.blind_no_room = true;
		}
		return RSX_OK;
	}
	RSX_TRY(seg_layout<KT>(c, n));   // (Sc2Cfg<KT, NoVal> and <KT, VT> have the same tile: 32 Ki elements)
	static_assert((int)C2::TILE == (int)Sc2Cfg<KT, NoVal>::TILE, "one layout for both");
	RSX_TRY(c.gscan.ensure(256 * sizeof(u64)));
	const u64 ntiles0 = (n + C2::TILE - 1) / C2::TILE;
	const u64 rows = ntiles0 + 256;
	const size_t st_bytes = 256 + rows * 256 * 4;
	SegCtl *ctl = (SegCtl *)c.seg.p;
	SegTile *tiles = (SegTile *)((char *)c.seg.p + c.seg_tiles_off);
	LeafSeg *segtab = (LeafSeg *)((char *)c.seg.p + c.seg_segtab_off);
	u32 *btile = (u32 *)((char *)c.seg.p + c.seg_btile_off);
	u64 *off1 = (u64 *)c.gscan.p;
	if (!c.seg_ev)
		HIP_TRY(hipEventCreateWithFlags(&c.seg_ev, hipEventDisableTiming));
	c.host_segctl->mode = SEG_MODE_NONE;
	RSX_TRY(blind_forget_device_backoff(c));
	hipLaunchKernelGGL((rsx_blind_precheck_kernel<KT>), dim3(1 + 512), dim3(1024), 0, c.stream, kin, (u64)n, ka, ctl, c.plan(),
	                   c.dev_host_plan, (u32x4 *)((char *)c.seg.p + c.seg_status_off), (u64)(2 * st_bytes / 16),
	                   (u32)sizeof(KT),   // (every column kept: the callers' parity rule below counts on it)
	                   0u, 0u,
	                   // rank sorts (no keys wanted back): keys whose byte columns do not spread but whose VARYING bits would, packed
	                   // together, go by those (SegCtl::compact, README.md:716-758)
	                   (u32)((vin == nullptr && kfinal == nullptr && !env().no_packed_keys) ? 1 : 0), (u32)(g_in_async ? 1 : 0));
	SegArgs sa{};
	sa.ctl = ctl;
	sa.hist = (const u32 *)((char *)c.seg.p + c.seg_hist_off);
	sa.tiles = tiles;
	sa.slots = (u32)sizeof(KT) - 1;
	sa.overflow = &ctl->overflow;
	char *base0 = (char *)c.seg.p + c.seg_status_off, *base1 = base0 + st_bytes;
	// the two parts of the level-1 slots (SegArgs): one base per array for the level-1 pass's stores and the parts' offsets from it;
	// for the level-2 pass the spare buffer as the array and the scratch part's virtual slot 0 as the other one
	KT *k1out = (KT *)c.slack1.p, *k1lo = (KT *)c.slack1.p;
	VT *v1out = (VT *)c.slack1_v.p, *v1lo = (VT *)c.slack1_v.p;
	const void *k1hi = nullptr, *v1hi = nullptr;
	sa.lo_slots = lo;
	if (klo) {
		const uintptr_t lo_a = (uintptr_t)kspare, hi_a = (uintptr_t)c.slack1.p - (size_t)lo * cap1 * sizeof(KT), base_a = std::min(lo_a, hi_a);
		sa.out_off_lo = (u32)((lo_a - base_a) / sizeof(KT));
		sa.out_off_hi = (u32)((hi_a - base_a) / sizeof(KT));
		k1out = (KT *)base_a;
		k1lo = kspare;
		k1hi = (const void *)hi_a;
	}
	if (vlo) {
		const uintptr_t lo_a = (uintptr_t)vspare, hi_a = (uintptr_t)c.slack1_v.p - (size_t)lo * cap1 * sizeof(VT), base_a = std::min(lo_a, hi_a);
		sa.v_off_lo = (u32)((lo_a - base_a) / sizeof(VT));
		sa.v_off_hi = (u32)((hi_a - base_a) / sizeof(VT));
		v1out = (VT *)base_a;
		v1lo = vspare;
		v1hi = (const void *)hi_a;
	}
	{
		ProfScope prof(1, (u64)n * (2 * sizeof(KT) + (vin ? 2 : 1) * sizeof(VT)), c.stream);
		sa.slack_cap = cap1;
		const u32 flags = (u32)SCATTER_SEG_SLACK | (u32)SCATTER_BLIND | (u32)SCATTER_BLIND_TOP | (vin ? 0u : (u32)SCATTER_GEN_INDEX);
		hipLaunchKernelGGL((rsx_scatter2_kernel<KT, VT, u32, C2, false, DIG_GENERIC, false, KT, true>), dim3((unsigned)ntiles0),
		                   dim3(C2::BLOCK), 0, c.stream, kin, k1out, vin, v1out, (u64)n, 0u,
		                   (const u64 *)c.ghist(), 1u, (u32 *)(base1 + 256), (u32 *)base1, ka, flags, (u64 *)nullptr,
		                   (const Plan *)c.plan(), 0u, 0u, (const u32 *)nullptr, sa);
	}
	sa.out_off_lo = sa.out_off_hi = sa.v_off_lo = sa.v_off_hi = 0;
	sa.kin_hi = k1hi;
	sa.vin_hi = v1hi;
	hipLaunchKernelGGL(rsx_seg_tiles_kernel, dim3(32), dim3(256), 0, c.stream, (const u64 *)c.ghist(), (u64)n, (const Plan *)c.plan(),
	                   (u32)C2::TILE, tiles, ctl, btile, off1, cap1, (const u32 *)(base1 + 256), (u32)ntiles0);
	{
		ProfScope prof(1, (u64)n * (sizeof(KT) + sizeof(K2) + 2 * sizeof(VT)), c.stream);
		sa.slack_cap = cap2;
		hipLaunchKernelGGL((rsx_scatter2_kernel<KT, VT, u32, C2, false, DIG_GENERIC, false, K2, true>), dim3((unsigned)rows),
		                   dim3(C2::BLOCK), 0, c.stream, (const KT *)k1lo, (K2 *)c.slack.p, (const VT *)v1lo,
		                   (VT *)c.slack_v.p, (u64)n, 0u, (const u64 *)c.ghist(), 1u, (u32 *)(base0 + 256), (u32 *)base0, ka,
		                   (u32)SCATTER_SEG_SLACK | (u32)SCATTER_BLIND, (u64 *)nullptr, (const Plan *)c.plan(), 0u, 0u,
		                   (const u32 *)nullptr, sa);
	}
	hipLaunchKernelGGL((rsx_seg_slack_plan_kernel<u32>), dim3(256), dim3(256), 0, c.stream, (const u32 *)(base0 + 256), (const u32 *)btile,
	                   (const u64 *)c.ghist(), (const Plan *)c.plan(), ctl, segtab, cap2, c.dev_host_segctl, (const u64 *)off1,
	                   1u | ((env().probe & 1u) << 8));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c.seg_ev, c.stream));
	{
		ProfScope prof(2, (u64)n * (sizeof(K2) + 2 * sizeof(VT) + (kfinal ? sizeof(KT) : 0)), c.stream);
		if (!env().no_leaf16) {
			// the compounds (key half, position) through one placement and the register passes (rsx_leafp_kernel); what it
			// leaves alone -- or everything, if the sample saw the keys' low bits cluster -- through the LDS passes of round 3
			// Three shapes by the slots' capacity (the host knows it): a leaf's fixed costs -- cells zeroed and scanned, barriers
			// of the whole workgroup -- follow the shape, not the pairs in it (16 Mi pairs through the 5120-pair shape: 0.48 ms
			// for the leaves alone, as much as 128 Mi pairs take)
			typedef LeafKCfg<512, 5120, 8> P5;
			typedef LeafKCfg<256, 2560, 8, 11> P2;
			typedef LeafKCfg<128, 1280, 6, 10> P1;
			u32 *redo = (u32 *)((char *)c.seg.p + c.seg_redo_off);
#define RSX_LEAFP(P) \
	hipLaunchKernelGGL((rsx_leafp_kernel<KT, VT, P, true>), dim3(65536u), dim3(P::BLOCK), 0, c.stream, (const KT *)c.slack.p, \
	                   (const VT *)c.slack_v.p, cap2, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab, ctl, ka, redo, \
	                   (u32)env().leaf16_maxbin)
			typedef LeafKCfg<64, 256, 8, 9> P0;    // slots of up to 256 pairs: a wave per leaf
			typedef LeafKCfg<64, 512, 8, 10> P0b;  // ... and of 512 (arrays of 11.5 .. 27 Mi pairs)
			if (big)
				RSX_LEAFP(P10);
			else if (cap2 <= (u32)P0::CAP && !env().no_leaf16q)
				RSX_LEAFP(P0);
			else if (cap2 <= (u32)P0b::CAP && !env().no_leaf16q)
				RSX_LEAFP(P0b);
			else if (cap2 <= (u32)P1::CAP)
				RSX_LEAFP(P1);
			else if (cap2 <= (u32)P2::CAP)
				RSX_LEAFP(P2);
			else
				RSX_LEAFP(P5);
#undef RSX_LEAFP
			if (big)
				hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, LB, true>), dim3(4096), dim3(LB::BLOCK), 0, c.stream, (const KT *)c.slack.p,
				                   (const VT *)c.slack_v.p, cap2, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab,
				                   (const SegCtl *)ctl, ka, (u32)HYB_TWO_LEVEL, (const u64 *)nullptr, (u64)0, (const u32 *)redo);
			else
				hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, L, true>), dim3(4096), dim3(L::BLOCK), 0, c.stream, (const KT *)c.slack.p,
				                   (const VT *)c.slack_v.p, cap2, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab,
				                   (const SegCtl *)ctl, ka, (u32)HYB_TWO_LEVEL, (const u64 *)nullptr, (u64)0, (const u32 *)redo);
		} else {
			if (big)
				hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, LB, true>), dim3(env().leaf_grid), dim3(LB::BLOCK), 0, c.stream, (const KT *)c.slack.p,
				                   (const VT *)c.slack_v.p, cap2, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab,
				                   (const SegCtl *)ctl, ka);
			else
				hipLaunchKernelGGL((rsx_leaf_pairs_kernel<KT, VT, L, true>), dim3(env().leaf_grid), dim3(L::BLOCK), 0, c.stream, (const KT *)c.slack.p,
				                   (const VT *)c.slack_v.p, cap2, kfinal, vfinal, (const Plan *)c.plan(), (const LeafSeg *)segtab,
				                   (const SegCtl *)ctl, ka);
		}
	}
	HIP_TRY(hipGetLastError());
	*enqueued = 1;
	return RSX_OK;
}

// ... and the blocking sorts' use of it: the host waits for the verdict (the event lies behind the slack plan kernel, in front of
// the leaves) and remembers an attempt that was called off (blind_called_off: back-off)
template <typename KT, typename VT>
int pairs_blind(Ctx &c, const KT *kin, const VT *vin, KT *kfinal, VT *vfinal, size_t n, KdfArgs<KT> ka, rsx_info *info, int *done,
                KT *kspare = nullptr, VT *vspare = nullptr)
{
	*done = 0;
	int enqueued = 0;
	const size_t pmark = prof_mark();
	RSX_TRY((pairs_blind_enqueue<KT, VT>(c, kin, vin, kfinal, vfinal, n, ka, &enqueued, kspare, vspare)));
	if (!enqueued)
		return RSX_OK;
	HIP_TRY(hipEventSynchronize(c.seg_ev));
	if (c.host_segctl->mode != SEG_MODE_LEAVES) {
		blind_called_off(c, blind_kind<KT>(sizeof(VT), vin == nullptr));   // (no payloads given: a rank sort)
		prof_called_off(pmark, c.stream);
		return RSX_OK;
	}
	c.blind_backoff[blind_kind<KT>(sizeof(VT), vin == nullptr)] = 0;
	info_from_plan(info, *c.host_plan);
	if (info)
		info->hybrid = 5u;
	*done = 1;
	return RSX_OK;
}

// ---- key + payload -----------------------------------------------------------------
template <typename KT, typename VT>
int sort_pairs_device_impl(Ctx &c, KT *k0, KT *k1, VT *v0, VT *v1, size_t n, int dtype, int order, rsx_info *info);

// RSX_VERIFY=2 (as for keys-only sorts): the sort on its usual route, bracketed by checksums of the PAIRS: the result's keys
// must not descend and its key sum and pair mix must be the input's.
template <typename KT, typename VT>
int sort_pairs_device(Ctx &c, KT *k0, KT *k1, VT *v0, VT *v1, size_t n, int dtype, int order, rsx_info *info)
{
	if (!env().verify_whole)
		return sort_pairs_device_impl<KT, VT>(c, k0, k1, v0, v1, n, dtype, order, info);
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	RSX_TRY(c.vsum.ensure(6 * sizeof(u64)));
	u64 *vs = (u64 *)c.vsum.p;
	HIP_TRY(hipMemsetAsync(vs, 0, 6 * sizeof(u64), c.stream));
	hipLaunchKernelGGL((rsx_checksum_pairs_kernel<KT, VT>), dim3(2048), dim3(256), 0, c.stream, (const KT *)k0, (const VT *)v0, (u64)n, ka, vs);
	HIP_TRY(hipGetLastError());
	rsx_info local;
	memset(&local, 0, sizeof(local));
	rsx_info *inf = info ? info : &local;
	RSX_TRY((sort_pairs_device_impl<KT, VT>(c, k0, k1, v0, v1, n, dtype, order, inf)));
	const KT *kr = inf->result_in_aux ? k1 : k0;
	const VT *vr = inf->result_in_aux ? v1 : v0;
	hipLaunchKernelGGL((rsx_checksum_pairs_kernel<KT, VT>), dim3(2048), dim3(256), 0, c.stream, kr, vr, (u64)n, ka, vs + 3);
	HIP_TRY(hipGetLastError());
	u64 h[6];
	HIP_TRY(hipMemcpyAsync(h, vs, sizeof h, hipMemcpyDeviceToHost, c.stream));
	HIP_TRY(hipStreamSynchronize(c.stream));
	if (env().verify_inject)
		h[5] ^= 1;
	if (h[3] != 0 || h[1] != h[4] || h[2] != h[5])
		return fail(RSX_EVERIFY, "RSX_VERIFY=2: the result of a key + payload sort of %zu pairs (route %u) is %s: %llu descents, key sum %s, pair mix %s",
		            n, inf->hybrid, h[3] ? "not sorted" : "not a permutation of the input's pairs", (unsigned long long)h[3],
		            h[1] == h[4] ? "kept" : "changed", h[2] == h[5] ? "kept" : "changed");
	return RSX_OK;
}

template <typename KT, typename VT>
int sort_pairs_device_impl(Ctx &c, KT *k0, KT *k1, VT *v0, VT *v1, size_t n, int dtype, int order, rsx_info *info)
{
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (c.fast && n * 2 * (sizeof(KT) + sizeof(VT)) <= SMALL_PAIR_BYTES && !env().no_small_sort && !capture_armed()) {
		ProfScope prof(1, (u64)n * 2 * (sizeof(KT) + sizeof(VT)), c.stream);
		hipLaunchKernelGGL((rsx_small_pairs_kernel<KT, VT, false>), dim3(1), dim3(1024), 0, c.stream, (const KT *)k0, k1, v0, v1,
		                   (u32)n, ka, c.dev_host_plan);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(c.stream));
		const Plan p = *c.host_plan;
		info_from_plan(info, p);
		if (info) {
			info->early_exit = p.sorted ? 2 : 0;
			info->result_in_aux = p.ncols & 1;
		}
		return RSX_OK;
	}
	if constexpr (sizeof(KT) == 4 && sizeof(VT) == 4) {
		if (blind_wanted<KT>(c, n, sizeof(VT))) {
			// all sizeof(KT) columns kept (pairs_blind: the sample proves it): the result lies where an even number of passes ends
			int done = 0;
			RSX_TRY((pairs_blind<KT, VT>(c, k0, v0, k0, v0, n, ka, info, &done, k1, v1)));
			if (done) {
				if (info)
					info->result_in_aux = 0;
				return RSX_OK;
			}
		}
	}
	Plan plan;
	RSX_TRY(plan_phase<KT>(c, k0, n, ka, &plan, 0, (c.fast && !capture_armed() && !verify_mode()) ? hybrid_caps_pairs<KT>(n, sizeof(VT)) : HybCaps{0, 0, 0, 0}));
	info_from_plan(info, plan);
	RSX_TRY(capture_hist(c, n, sizeof(KT)));
	if (plan.sorted) {
		if (info) {
			info->early_exit = 2;
			info->ncols = 0;
		}
		return RSX_OK;
	}
	if constexpr (sizeof(KT) == 4 && sizeof(VT) == 4) {
		if (plan.hyb == HYB_ONE_LEVEL) {
			// one MSB pass and the pairs' leaves (mid-size arrays)
			const u32 top = plan.cols[plan.ncols - 1];
			RSX_TRY((scatter_pass<KT, VT>(c, k0, k1, v0, v1, n, 8 * top, c.ghist() + 256 * top, ka, 0u)));
			const bool in_aux = (plan.ncols & 1) != 0;
			RSX_TRY((pairs_one_level<KT, VT>(c, k1, v1, in_aux ? k1 : k0, in_aux ? v1 : v0, n, ka)));
			if (info) {
				info->result_in_aux = in_aux;
				info->hybrid = 1;
			}
			return RSX_OK;
		}
		if (plan.hyb == HYB_TWO_LEVEL) {
			// two MSB passes (the second into slots) and leaves; on a slot's overflow: one pass per column, from (k0, v0) again
			const u32 top = plan.cols[plan.ncols - 1];
			RSX_TRY((scatter_pass<KT, VT>(c, k0, k1, v0, v1, n, 8 * top, c.ghist() + 256 * top, ka, 0u)));
			const bool in_aux = (plan.ncols & 1) != 0;
			bool ok = false;
			RSX_TRY((pairs_two_level<KT, VT>(c, k1, v1, in_aux ? k1 : k0, in_aux ? v1 : v0, n, ka, &ok)));
			if (ok) {
				if (info) {
					info->result_in_aux = in_aux;
					info->hybrid = 4;
				}
				return RSX_OK;
			}
		}
	}
	KT *kc = k0, *ko = k1;
	VT *vc = v0, *vo = v1;
	for (u32 i = 0; i < plan.ncols; ++i) {
		const u32 col = plan.cols[i];
		RSX_TRY((scatter_pass<KT, VT>(c, kc, ko, vc, vo, n, 8 * col, c.ghist() + 256 * col, ka, hot_flags(plan.hot, col))));
		std::swap(kc, ko);
		std::swap(vc, vo);
	}
	if (info)
		info->result_in_aux = kc == k1;
	return RSX_OK;
}

// ---- rank (stable argsort) ----------------------------------------------------------
// index halves H0 = ib, H1 = ib + n ping-pong exactly as radix_sort_rank.hpp:77-89;
// the keys travel with the indices through two workspace
// buffers instead of being gathered through the index as Listing 6 does.
// (a key type no wider than KT: keeps the instantiations of impossible combinations out of the build)
template <typename KT, typename N> using NarrowerOr = typename std::conditional<(sizeof(N) < sizeof(KT)), N, KT>::type;

// one pass of a rank sort over keys currently held as KCUR (the raw KT keys until the first narrowing, KDF-applied after)
template <typename KCUR, typename KT, typename IT>
int rank_pass(Ctx &c, const void *kin, void *kout, u32 out_bytes, const IT *vin, IT *vout, size_t n, u32 shift, const u64 *gbase,
              const KdfArgs<KT> &ka0, bool applied, u32 flags, u32 oshift)
{
	KdfArgs<KCUR> ka{0, 0, 0};
	if constexpr (std::is_same<KCUR, KT>::value)
		if (!applied)
			ka = ka0;
	return scatter_pass_to<KCUR, IT>(c, (const KCUR *)kin, kout, out_bytes, vin, vout, n, shift, gbase, ka, flags, oshift);
}

// the runs of contiguous set bits of `mask`, lowest first, packed towards bit 0; false if there are more than eight
bool bit_runs(u64 mask, BitRuns *out)
{
	out->n = 0;
	u32 dst = 0;
	for (u32 b = 0; b < 64;) {
		if (!((mask >> b) & 1)) {
			++b;
			continue;
		}
		u32 e = b;
		while (e < 64 && ((mask >> e) & 1))
			++e;
		if (out->n == 8)
			return false;
		out->src[out->n] = (uint8_t)b;
		out->len[out->n] = (uint8_t)(e - b);
		out->dst[out->n] = (uint8_t)dst;
		dst += e - b;
		++out->n;
		b = e;
	}
	return true;
}

// want_half: -1 = the half the number of kept columns dictates (radix_sort_rank.hpp:91); 0 / 1 = leave the ranks in that half
// whatever the number of passes is (the first pass generates its indices, so it can write to either half).
template <typename KT, typename IT>
int sort_rank_device_impl(Ctx &c, const KT *src, IT *ib, size_t n, int dtype, int order, void **result, rsx_info *info, int want_half);

// RSX_VERIFY=2: the ranks must be a permutation of 0 .. n-1 (sum and mix) through which the keys do not descend, equal keys in
// index order (radix_sort_rank.hpp:82-90: stable) -- checked on the device, whatever route the sort took.
template <typename KT, typename IT>
int sort_rank_device(Ctx &c, const KT *src, IT *ib, size_t n, int dtype, int order, void **result, rsx_info *info, int want_half = -1)
{
	RSX_TRY((sort_rank_device_impl<KT, IT>(c, src, ib, n, dtype, order, result, info, want_half)));
	if (!env().verify_whole || n < 2)
		return RSX_OK;
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	RSX_TRY(c.vsum.ensure(6 * sizeof(u64)));
	u64 *vs = (u64 *)c.vsum.p;
	HIP_TRY(hipMemsetAsync(vs, 0, 6 * sizeof(u64), c.stream));
	hipLaunchKernelGGL((rsx_check_ranks_kernel<KT, IT>), dim3(2048), dim3(256), 0, c.stream, (const KT *)nullptr, (const IT *)nullptr, (u64)n, ka, vs);
	hipLaunchKernelGGL((rsx_check_ranks_kernel<KT, IT>), dim3(2048), dim3(256), 0, c.stream, src, (const IT *)*result, (u64)n, ka, vs + 3);
	HIP_TRY(hipGetLastError());
	u64 h[6];
	HIP_TRY(hipMemcpyAsync(h, vs, sizeof h, hipMemcpyDeviceToHost, c.stream));
	HIP_TRY(hipStreamSynchronize(c.stream));
	if (env().verify_inject)
		h[5] ^= 1;
	if (h[3] != 0 || h[1] != h[4] || h[2] != h[5])
		return fail(RSX_EVERIFY, "RSX_VERIFY=2: the ranks of a sort of %zu keys (route %u) are %s: %llu places out of order, rank sum %s, rank mix %s",
		            n, info ? info->hybrid : 0u, h[3] ? "not the stable order" : "not a permutation of 0 .. n-1", (unsigned long long)h[3],
		            h[1] == h[4] ? "right" : "wrong", h[2] == h[5] ? "right" : "wrong");
	return RSX_OK;
}

template <typename KT, typename IT>
int sort_rank_device_impl(Ctx &c, const KT *src, IT *ib, size_t n, int dtype, int order, void **result, rsx_info *info, int want_half)
{
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (c.fast && n * 2 * (sizeof(KT) + sizeof(IT)) <= SMALL_PAIR_BYTES && !env().no_small_sort && !capture_armed() &&
	    want_half < 0) {
		ProfScope prof(1, (u64)n * (sizeof(KT) + 2 * sizeof(IT)), c.stream);
		hipLaunchKernelGGL((rsx_small_pairs_kernel<KT, IT, true>), dim3(1), dim3(1024), 0, c.stream, src, (KT *)nullptr, ib, ib + n,
		                   (u32)n, ka, c.dev_host_plan);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(c.stream));
		const Plan p = *c.host_plan;
		info_from_plan(info, p);
		if (info) {
			info->early_exit = p.sorted ? 2 : 0;
			info->result_in_aux = p.ncols & 1;
		}
		*result = (p.ncols & 1) ? ib + n : ib;   // radix_sort_rank.hpp:91 (sorted: first half = iota)
		return RSX_OK;
	}
	if constexpr (sizeof(KT) == 4 && sizeof(IT) == 4) {
		if (want_half < 0 && !env().compact_bits && blind_wanted<KT>(c, n, sizeof(IT), true)) {
			int done = 0;
			// (spare buffers: the index buffer's two halves -- the second one, which the reference's passes ping-pong through
			// (radix_sort_rank.hpp:77-91), for the keys' level-1 slots)
			RSX_TRY((pairs_blind<KT, IT>(c, src, (const IT *)nullptr, (KT *)nullptr, ib, n, ka, info, &done, (KT *)(ib + n), ib)));
			if (done) {   // four kept columns: the ranks are in the first half (radix_sort_rank.hpp:91)
				*result = ib;
				if (info)
					info->result_in_aux = 0;
				return RSX_OK;
			}
		}
	}
	Plan plan;
	RSX_TRY(plan_phase<KT>(c, src, n, ka, &plan, 0,
	                       (c.fast && want_half < 0 && !capture_armed() && !verify_mode() && !env().compact_bits)
	                           ? hybrid_caps_pairs<KT>(n, sizeof(IT)) : HybCaps{0, 0, 0, 0}));
	info_from_plan(info, plan);
	RSX_TRY(capture_hist(c, n, sizeof(KT)));
	if constexpr (sizeof(KT) == 4 && sizeof(IT) == 4) {
		if (!plan.sorted && plan.hyb == HYB_ONE_LEVEL && want_half < 0) {
			// mid-size arrays: one MSB pass of (key, index), which makes the indices, and the pairs' leaves write the ranks
			const u32 P = plan.ncols, top = plan.cols[P - 1];
			RSX_TRY(c.keys[0].ensure(n * sizeof(KT)));
			IT *fin = (P & 1) ? ib + n : ib, *scratch = (P & 1) ? ib : ib + n;
			RSX_TRY((scatter_pass<KT, IT>(c, src, (KT *)c.keys[0].p, (const IT *)fin, scratch, n, 8 * top, c.ghist() + 256 * top, ka,
			                              (u32)SCATTER_GEN_INDEX)));
			RSX_TRY((pairs_one_level<KT, IT>(c, (const KT *)c.keys[0].p, (const IT *)scratch, (KT *)nullptr, fin, n, ka)));
			*result = fin;
			if (info) {
				info->result_in_aux = fin != ib;
				info->hybrid = 1;
			}
			return RSX_OK;
		}
		if (!plan.sorted && plan.hyb == HYB_TWO_LEVEL && want_half < 0) {
			// Keys spread over their top two columns (cfg 4 (i)): two MSB passes of (key, index) -- the first makes the indices,
			// the second goes into slots -- and leaves that write the ranks where the parity rule says (radix_sort_rank.hpp:91).
			// On a slot's overflow nothing has been written to that half: the ordinary passes follow.
			const u32 P = plan.ncols, top = plan.cols[P - 1];
			RSX_TRY(c.keys[0].ensure(n * sizeof(KT)));
			IT *fin = (P & 1) ? ib + n : ib, *scratch = (P & 1) ? ib : ib + n;
			RSX_TRY((scatter_pass<KT, IT>(c, src, (KT *)c.keys[0].p, (const IT *)fin, scratch, n, 8 * top, c.ghist() + 256 * top, ka,
			                              (u32)SCATTER_GEN_INDEX)));
			bool ok = false;
			RSX_TRY((pairs_two_level<KT, IT>(c, (const KT *)c.keys[0].p, (const IT *)scratch, (KT *)nullptr, fin, n, ka, &ok)));
			if (ok) {
				*result = fin;
				if (info) {
					info->result_in_aux = fin != ib;
					info->hybrid = 4;
				}
				return RSX_OK;
			}
		}
	}
	if (plan.sorted) {                       // radix_sort_rank.hpp:52,:55-57: first half = iota
		hipLaunchKernelGGL((rsx_iota_kernel<IT>), dim3(1024), dim3(256), 0, c.stream, ib, (u64)n);
		HIP_TRY(hipGetLastError());
		if (info) {
			info->early_exit = 2;
			info->ncols = 0;
		}
		*result = ib;
		return RSX_OK;
	}
	const u32 P = plan.ncols;
	// README.md:716-758, "key compaction" (SURVEY.md 8 f4), behind RSX_COMPACT_BITS=1: when the bits that vary among the
	// keys (plan.vary, read off the histograms) fit fewer bytes than there are varying byte columns, the keys' varying bits
	// are packed together (one elementwise pass) and the packed values are rank-sorted instead: ceil(bits / 8) passes over
	// narrower keys.  The ranks are the same (all other bits are equal in every key) and they are left in the half the
	// reference's number of passes dictates.
	if (want_half < 0 && c.fast && sizeof(IT) == 4 && P > 1) {
		const bool compact_on = env().compact_bits;
		const u64 vary = ((u64)plan.vary_hi << 32) | plan.vary_lo;
		const u32 bits = (u32)__builtin_popcountll(vary), P2 = (bits + 7) / 8;
		BitRuns runs;
		if (compact_on && vary && P2 < P && bit_runs(vary, &runs)) {
			const u32 ob = P2 <= 1 ? 1 : P2 <= 2 ? 2 : P2 <= 4 ? 4 : 8;
			RSX_TRY(c.ckeys.ensure(n * ob));
			const dim3 grid(4096), block(256);
			rsx_info inner;
			int rc = RSX_EINVAL;
			const int half = (int)(P & 1);
			if (ob == 1) {
				hipLaunchKernelGGL((rsx_compact_bits_kernel<KT, uint8_t>), grid, block, 0, c.stream, src, (uint8_t *)c.ckeys.p, (u64)n, ka, runs);
				rc = sort_rank_device<uint8_t, IT>(c, (const uint8_t *)c.ckeys.p, ib, n, RSX_U8, RSX_ASCENDING, result, &inner, half);
			} else if (ob == 2) {
				hipLaunchKernelGGL((rsx_compact_bits_kernel<KT, uint16_t>), grid, block, 0, c.stream, src, (uint16_t *)c.ckeys.p, (u64)n, ka, runs);
				rc = sort_rank_device<uint16_t, IT>(c, (const uint16_t *)c.ckeys.p, ib, n, RSX_U16, RSX_ASCENDING, result, &inner, half);
			} else if (ob == 4) {
				hipLaunchKernelGGL((rsx_compact_bits_kernel<KT, u32>), grid, block, 0, c.stream, src, (u32 *)c.ckeys.p, (u64)n, ka, runs);
				rc = sort_rank_device<u32, IT>(c, (const u32 *)c.ckeys.p, ib, n, RSX_U32, RSX_ASCENDING, result, &inner, half);
			} else {
				hipLaunchKernelGGL((rsx_compact_bits_kernel<KT, u64>), grid, block, 0, c.stream, src, (u64 *)c.ckeys.p, (u64)n, ka, runs);
				rc = sort_rank_device<u64, IT>(c, (const u64 *)c.ckeys.p, ib, n, RSX_U64, RSX_ASCENDING, result, &inner, half);
			}
			RSX_TRY(rc);
			if (info)
				info->result_in_aux = P & 1;   // (ncols / cols stay the reference's: those of the original keys)
			return RSX_OK;
		}
	}
	if (P > 1) {
		RSX_TRY(c.keys[0].ensure(n * sizeof(KT)));
		if (P > 2)
			RSX_TRY(c.keys[1].ensure(n * sizeof(KT)));
	}
	// want_half: the halves swap roles when the number of passes would leave the ranks in the other one
	const bool swap_halves = want_half >= 0 && (int)(P & 1) != want_half;
	IT *H[2] = {swap_halves ? ib + n : ib, swap_halves ? ib : ib + n};
	if (c.fast && sizeof(IT) == 4 && !env().no_narrow_keys) {
		// The ranks are the only output, so a pass hands on just the key bytes that later passes look at: once those fit a
		// narrower type the keys are written as kdf(key) >> (8 * next column) in that type, and the passes after it read
		// it with the identity KDF.  `base_col`: the column that sits in the low byte of the current representation.
		const void *kin = src;
		u32 in_bytes = sizeof(KT), base_col = 0;
		bool applied = false;
		for (u32 i = 0; i < P; ++i) {
			const u32 col = plan.cols[i];
			u32 flags = hot_flags(plan.hot, col);
			if (i == 0)
				flags |= SCATTER_GEN_INDEX;
			if (i == P - 1)
				flags |= SCATTER_SKIP_KEYS;
			u32 out_bytes = in_bytes, oshift = 0, next = base_col;
			if (i + 1 < P) {
				const u32 need = (u32)sizeof(KT) - plan.cols[i + 1];   // bytes from the next kept column up
				const u32 fit = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
				if (fit < in_bytes) {
					out_bytes = fit;
					next = plan.cols[i + 1];
					oshift = 8 * (next - base_col);
				}
			}
			void *kout = c.keys[i & 1].p;
			const u32 shift = 8 * (col - base_col);
			const u64 *gb = c.ghist() + 256 * col;
			int rc = RSX_EINVAL;
			if (in_bytes == sizeof(KT))
				rc = rank_pass<KT, KT, IT>(c, kin, kout, out_bytes, H[i & 1], H[(i + 1) & 1], n, shift, gb, ka, applied, flags, oshift);
			else if (in_bytes == 4)
				rc = rank_pass<NarrowerOr<KT, u32>, KT, IT>(c, kin, kout, out_bytes, H[i & 1], H[(i + 1) & 1], n, shift, gb, ka, true,
				                                           flags, oshift);
			else if (in_bytes == 2)
				rc = rank_pass<NarrowerOr<KT, uint16_t>, KT, IT>(c, kin, kout, out_bytes, H[i & 1], H[(i + 1) & 1], n, shift, gb, ka,
				                                                true, flags, oshift);
			else if (in_bytes == 1)
				rc = rank_pass<uint8_t, KT, IT>(c, kin, kout, out_bytes, H[i & 1], H[(i + 1) & 1], n, shift, gb, ka, true, flags,
				                                oshift);
			RSX_TRY(rc);
			if (out_bytes != in_bytes) {
				in_bytes = out_bytes;
				base_col = next;
				applied = true;
			}
			kin = kout;
		}
	} else
	for (u32 i = 0; i < P; ++i) {
		const u32 col = plan.cols[i];
		const KT *kin = i == 0 ? src : (const KT *)c.keys[(i - 1) & 1].p;
		KT *kout = (KT *)c.keys[i & 1].p;
		u32 flags = hot_flags(plan.hot, col);
		if (i == 0)
			flags |= SCATTER_GEN_INDEX;
		if (i == P - 1)
			flags |= SCATTER_SKIP_KEYS;
		RSX_TRY((scatter_pass<KT, IT>(c, kin, kout, H[i & 1], H[(i + 1) & 1], n, 8 * col, c.ghist() + 256 * col, ka, flags)));
	}
	*result = H[P & 1];                      // radix_sort_rank.hpp:91
	if (info)
		info->result_in_aux = *result != (void *)ib;
	return RSX_OK;
}

// ---- stable argsort, no host synchronisation: every pass is device-scheduled, the ranks always end in the FIRST half -----------
// radix_sort_rank.hpp:97-112 for callers that cannot wait (graphs, the multi-GPU chunk pipeline).  The number of passes is only
// known on the device, so the passes take their buffers from the plan (SCATTER_RANK_ASYNC): the last one writes the first
// half.  Keys travel at full width (which narrower type a pass could write is a choice of kernel, i.e. the host's).
template <typename KT, typename IT>
int sort_rank_inplace_async(Ctx &c, const KT *src, IT *ib, size_t n, int dtype, int order)
{
	if (!c.fast)
		return fail(RSX_EHIP, "rsx_sort_rank_inplace_async needs the fast scatter kernel (the device self-check failed on this device)");
	c.async_tried_blind = false;   // (rsx_async_route reports THIS call: set again below if an attempt is enqueued)
	c.async_small = false;
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	if (n * 2 * (sizeof(KT) + sizeof(IT)) <= SMALL_PAIR_BYTES) {
		hipLaunchKernelGGL((rsx_small_pairs_kernel<KT, IT, true>), dim3(1), dim3(1024), 0, c.stream, src, (KT *)nullptr, ib, ib + n,
		                   (u32)n, ka, c.dev_host_plan, true);
		HIP_TRY(hipGetLastError());
		c.async_small = true;
		return RSX_OK;
	}
	RSX_TRY(c.keys[0].ensure(n * sizeof(KT)));
	RSX_TRY(c.keys[1].ensure(n * sizeof(KT)));
	const size_t status_total = status_bytes<KT, IT>(n) * sizeof(KT);
	// the attempt without a histogram first (4-byte keys, 4-byte indices, 16 Mi .. 2^28 keys: its leaves write the ranks to the
	// first half), the histogram-first kernels behind it gated on its verdict -- as sort_keys_inplace_async
	int blind = 0;
	ProfAsyncVerdict pverdict(c.stream);
	if constexpr (sizeof(KT) == 4 && sizeof(IT) == 4) {
		if (async_pairs_blind_ok<KT>(c, n, sizeof(IT)))
			RSX_TRY((pairs_blind_enqueue<KT, IT>(c, src, (const IT *)nullptr, (KT *)nullptr, ib, n, ka, &blind, (KT *)(ib + n), ib)));
	}
	if (blind)
		pverdict.attempt_enqueued((const SegCtl *)c.seg.p);
	c.pass_gate = blind ? (const SegCtl *)c.seg.p : nullptr;
	c.async_tried_blind = blind != 0;
	int rc = plan_phase<KT>(c, src, n, ka, nullptr, status_total);
	c.pass_alt = c.keys[1].p;
	for (u32 i = 0; i < sizeof(KT) && rc == RSX_OK; ++i)   // pass i = the i-th kept column, if there is one
		rc = scatter_pass<KT, IT>(c, src, (KT *)c.keys[0].p, ib, ib + n, n, 0, c.ghist(), ka, SCATTER_RANK_ASYNC, c.plan(), (int)i, i);
	c.pass_alt = nullptr;
	c.pass_gate = nullptr;
	RSX_TRY(rc);
	pverdict.gated_enqueued();
	// sorted keys: no pass ran, the ranks are 0 .. n-1 (radix_sort_rank.hpp:52,:55-57)
	hipLaunchKernelGGL((rsx_iota_if_sorted_kernel<IT>), dim3(1024), dim3(256), 0, c.stream, ib, (u64)n, (const Plan *)c.plan());
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

bool is_device_ptr(const void *p)
{
	hipPointerAttribute_t attr;
	hipError_t e = hipPointerGetAttributes(&attr, p);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// dispatch on key width
// ---- one plain scatter pass by the top KDF byte (rsx_msd_split_device) ---------------------------------------------
template <typename KT>
int msd_split(Ctx &c, const KT *src, KT *dst, size_t n, int dtype, int order, u32 col, uint64_t *top_hist)
{
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	const u64 *top = c.ghist() + 256 * col;
	RSX_TRY(launch_hist<KT>(c, src, n, ka, c.ghist(), c.unsorted(), 1u << col));   // only the column split by
	HIP_TRY(hipMemcpyAsync(c.host_hist, top, 256 * sizeof(u64), hipMemcpyDeviceToHost, c.stream));   // counts, before the scan
	hipLaunchKernelGGL((rsx_plan_kernel<KT>), dim3(sizeof(KT)), dim3(256), 0, c.stream, src, (u64)n, c.ghist(), ka, c.kept(),
	                   c.hotd());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c.stream));
	u64 total = 0;
	u64 most = 0;
	for (int i = 0; i < 256; ++i) {
		top_hist[i] = c.host_hist[i];
		total += top_hist[i];
		most = std::max<u64>(most, top_hist[i]);
	}
	if (total != n)
		return fail(RSX_EHIP, "rsx_msd_split_device: digit counts sum to %llu, n = %zu", (unsigned long long)total, n);
	const u32 flags = most >= (u64)n / 8 + 1 ? hot_flags(1u << col, col) : 0u;   // as Plan::hot (rsx_plan_kernel)
	return scatter_pass<KT, NoVal>(c, src, dst, nullptr, nullptr, n, 8 * col, top, ka, flags);
}

// The same pass for a caller that HAS the shard's column counts (rsx_histogram_device: one read gave every column): nothing
// is counted again and nothing waits for the host.
template <typename KT>
int msd_split_known(Ctx &c, const KT *src, KT *dst, size_t n, int dtype, int order, u32 col, const u64 *d_counts, bool hot)
{
	const KdfArgs<KT> ka = make_kdf<KT>(dtype, order);
	HIP_TRY(hipMemcpyAsync(c.ghist(), d_counts, sizeof(KT) * 256 * sizeof(u64), hipMemcpyDeviceToDevice, c.stream));
	hipLaunchKernelGGL((rsx_plan_kernel<KT>), dim3(sizeof(KT)), dim3(256), 0, c.stream, src, (u64)n, c.ghist(), ka, c.kept(),
	                   c.hotd());   // (the exclusive scans, radix_sort.hpp:72-80)
	HIP_TRY(hipGetLastError());
	// (a dominant digit -- the caller has the counts on the host and says so: the ballot-ranked kernel, as msd_split's;
	// its hot digits are rsx_plan_kernel's, on the device)
	return scatter_pass<KT, NoVal>(c, src, dst, nullptr, nullptr, n, 8 * col, c.ghist() + 256 * col, ka,
	                               hot ? hot_flags(1u << col, col) : 0u);
}

#include "rsx_multi_state.hpp"   // rsx_sort_multi: per-rank streams, buffers, phases, peer access

#define RSX_DISPATCH_KT(dtype, CALL)                               \
	switch (dtype_size(dtype)) {                                   \
	case 1: { typedef uint8_t KT; CALL; } break;                   \
	case 2: { typedef uint16_t KT; CALL; } break;                  \
	case 4: { typedef uint32_t KT; CALL; } break;                  \
	case 8: { typedef u64 KT; CALL; } break;                       \
	default: return fail(RSX_EINVAL, "unknown dtype %d", (int)(dtype)); \
	}

}  // namespace

// =================================================================================
// extern "C" surface
// =================================================================================

extern "C" {

int rsx_device_count(void)
{
	std::lock_guard<std::mutex> lock(g_mu);
	return probe_devices();
}

const char *rsx_last_error(void) { return g_err; }
const char *rsx_version(void) { return "rsx 0.1 (gfx950)"; }
size_t rsx_dtype_size(rsx_dtype dtype) { return dtype_size(dtype); }

size_t rsx_workspace_bytes(size_t n, rsx_dtype dtype, size_t payload_bytes)
{
	const size_t kb = dtype_size(dtype);
	if (!kb)
		return 0;
	// upper bound over the kernels' shapes: quarter tiles (8 Ki elements, 4 Ki with an 8-byte element), a region of status
	// words per pass, the histogram kernel's rows (at most 512 workgroups)
	const size_t elem = kb > payload_bytes ? kb : payload_bytes;
	const size_t tile = elem == 8 ? 4096 : 8192;
	const size_t tiles = (n + tile - 1) / tile;
	const size_t status = (256 + tiles * 256 * (n >= (1ull << 30) ? 8 : 4)) * kb;
	return 512 + kb * 256 * 8 + ((status + 255) & ~(size_t)255) + (size_t)512 * kb * 256 * 4 + 256;
}

void rsx_release(void)
{
	std::lock_guard<std::mutex> lock(g_mu);
	for (auto &kv : g_ctx) {
		(void)hipSetDevice(kv.first.first);
		kv.second->release();
		delete kv.second;
	}
	g_ctx.clear();
	for (auto &kv : g_multi_streams) {
		(void)hipSetDevice(kv.first.first);
		(void)hipStreamDestroy(kv.second);
	}
	g_multi_streams.clear();
	for (auto &kv : g_multi_bufs) {
		(void)hipSetDevice(kv.first.first);
		kv.second.shard.release();
		kv.second.part.release();
		kv.second.recv.release();
		kv.second.aux.release();
		kv.second.misc.release();
	}
	g_multi_bufs.clear();
}

namespace {

// A context whose device state lies in the caller's workspace: [flags 256][plan 64 + pad][histogram][status regions]
// [histogram rows].  Everything a captured graph of the *_ws entry points refers to is inside that workspace.
int borrow_ctx(Ctx &v, void *stream, void *ws, size_t ws_bytes, size_t n, size_t kb, size_t status_total, char **end = nullptr)
{
	// (no context of the library's own is created or touched: the call may be inside a stream capture, where nothing may
	// be allocated; the device self-check has run when any other entry point was used before, otherwise it runs now)
	int dev = 0;
	{
		std::lock_guard<std::mutex> lock(g_mu);
		if (probe_devices() <= 0)
			return fail(RSX_ENODEVICE, "no gfx950 (MI355X) device visible to HIP; this library has no CPU path");
		HIP_TRY(hipGetDevice(&dev));
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		if (g_lds_order_ok.find(dev) == g_lds_order_ok.end() && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess &&
		    cap != hipStreamCaptureStatusNone)
			return fail(RSX_EINVAL, "the library's first use on this device is inside a stream capture: call rsx_device_count() "
			                        "and any sort (or rsx_sort_inplace_async_ws itself) once before capturing -- the device "
			                        "self-check cannot run inside a capture");
		(void)hipGetLastError();
		if (!lds_order_selfcheck(dev))
			return fail(RSX_EHIP, "the device-scheduled sorts need the fast scatter kernel (the device self-check failed on this device)");
	}
	if (((uintptr_t)ws & 255) != 0)
		return fail(RSX_EINVAL, "the workspace must be 256-byte aligned");
	const size_t hist_bytes = kb * 256 * sizeof(u64), hpart_bytes = (size_t)512 * kb * 256 * sizeof(u32);
	const size_t need = 512 + hist_bytes + ((status_total + 255) & ~(size_t)255) + hpart_bytes;
	if (!ws || ws_bytes < need)
		return fail(RSX_EINVAL, "workspace of %zu bytes, %zu needed for %zu keys (rsx_workspace_bytes gives an upper bound)", ws_bytes, need, n);
	char *p = (char *)ws;
	v.device = dev;
	v.stream = (hipStream_t)stream;
	v.fast = true;
	v.small.borrow(p, 256);
	v.dev_host_plan = (Plan *)(p + 256);          // (the kernels' second copy of the plan: nobody reads it on the host)
	v.host_plan = nullptr;
	p += 512;
	v.hist.borrow(p, hist_bytes);
	p += hist_bytes;
	v.status.borrow(p, (status_total + 255) & ~(size_t)255);
	p += (status_total + 255) & ~(size_t)255;
	v.hpart.borrow(p, hpart_bytes);
	if (end)
		*end = p + hpart_bytes;
	return RSX_OK;
}

}  // namespace

int rsx_capture_histogram(uint64_t *hist, size_t entries)
{
	g_capture_dst = (u64 *)hist;
	g_capture_entries = hist ? entries : 0;
	return RSX_OK;
}

int rsx_sort_inplace_async(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order, void *stream)
{
	if (!dtype_size(dtype) || (n && (!d_buf || !d_scratch)))
		return fail(RSX_EINVAL, "rsx_sort_inplace_async: bad argument");
	if (n < 2)
		return RSX_OK;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	AsyncScope async_scope((hipStream_t)stream);
	RSX_DISPATCH_KT(dtype, return sort_keys_inplace_async<KT>(*c, (KT *)d_buf, (KT *)d_scratch, n, dtype, order));
	return RSX_OK;
}

int rsx_sort_inplace_async_hint(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order, void *stream, uint32_t hints)
{
	if (!dtype_size(dtype) || (n && (!d_buf || !d_scratch)))
		return fail(RSX_EINVAL, "rsx_sort_inplace_async_hint: bad argument");
	if (n < 2)
		return RSX_OK;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	AsyncScope async_scope((hipStream_t)stream);
	struct HintScope {   // (the sample kernel of the attempt enqueued by this call reads them: blind_enqueue)
		Ctx &c;
		HintScope(Ctx &c_, u32 h) : c(c_) { c.hints = h; }
		~HintScope() { c.hints = 0; }
	} hint_scope(*c, (u32)hints);
	RSX_DISPATCH_KT(dtype, return sort_keys_inplace_async<KT>(*c, (KT *)d_buf, (KT *)d_scratch, n, dtype, order));
	return RSX_OK;
}

int rsx_sort_inplace_async_ws(void *d_buf, void *d_scratch, size_t n, rsx_dtype dtype, rsx_order order, void *d_workspace,
                              size_t workspace_bytes, void *stream)
{
	const size_t kb = dtype_size(dtype);
	if (!kb || (n && (!d_buf || !d_scratch)))
		return fail(RSX_EINVAL, "rsx_sort_inplace_async_ws: bad argument");
	if (n < 2)
		return RSX_OK;
	Ctx view;
	size_t status_total = 0;
	RSX_DISPATCH_KT(dtype, status_total = (status_bytes<KT, NoVal>(n) * sizeof(KT)));
	char *end = nullptr;
	RSX_TRY(borrow_ctx(view, stream, d_workspace, workspace_bytes, n, kb, status_total, &end));
	// a workspace sized by rsx_workspace_bytes_fast also holds the slots of a sort without a histogram: the attempt is made in it
	RSX_DISPATCH_KT(dtype, borrow_blind<KT>(view, end, (char *)d_workspace + workspace_bytes, n));
	AsyncScope async_scope((hipStream_t)stream);
	RSX_DISPATCH_KT(dtype, return sort_keys_inplace_async<KT>(view, (KT *)d_buf, (KT *)d_scratch, n, dtype, order));
	return RSX_OK;
}

size_t rsx_workspace_bytes_fast(size_t n, rsx_dtype dtype)
{
	const size_t kb = dtype_size(dtype);
	if (!kb)
		return 0;
	size_t g = 0, sg = 0, s1 = 0, s2 = 0;
	if (kb >= 4 && n >= ((size_t)1 << 22) && n < ((size_t)1 << 30)) {
		if (kb == 4)
			blind_sizes<u32>(n, &g, &sg, &s1, &s2);
		else
			blind_sizes<u64>(n, &g, &sg, &s1, &s2);
	}
	return ((rsx_workspace_bytes(n, dtype, 0) + 255) & ~(size_t)255) + 256 + g + sg + s1 + s2;
}

int rsx_async_route_ws(const void *d_workspace, size_t workspace_bytes, size_t n, rsx_dtype dtype, void *stream, uint32_t *route)
{
	const size_t kb = dtype_size(dtype);
	if (!route || !d_workspace || !kb)
		return fail(RSX_EINVAL, "rsx_async_route_ws: bad argument");
	*route = 0;
	HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	Plan plan;
	HIP_TRY(hipMemcpy(&plan, (const char *)d_workspace + 64, sizeof(plan), hipMemcpyDeviceToHost));
	// (the layout of borrow_ctx / borrow_blind: the control block of the attempt lies behind the histogram kernel's rows and gscan)
	size_t status_total = 0;
	RSX_DISPATCH_KT(dtype, status_total = (status_bytes<KT, NoVal>(n) * sizeof(KT)));
	const size_t minimal = 512 + kb * 256 * sizeof(u64) + ((status_total + 255) & ~(size_t)255) + (size_t)512 * kb * 256 * sizeof(u32);
	if (kb >= 4 && workspace_bytes >= rsx_workspace_bytes_fast(n, dtype) && n >= ((size_t)1 << 22) && n < ((size_t)1 << 30)) {
		const char *p = (const char *)(((uintptr_t)d_workspace + minimal + 255) & ~(uintptr_t)255) + 256 * sizeof(u64);
		SegCtl ctl;
		HIP_TRY(hipMemcpy(&ctl, p, sizeof(ctl), hipMemcpyDeviceToHost));
		if (ctl.mode == SEG_MODE_LEAVES && ctl.blind == BLIND_GO) {
			*route = 5;
			return RSX_OK;
		}
	}
	if (!plan.sorted && plan.hyb == HYB_ONE_LEVEL)
		*route = 1;
	return RSX_OK;
}

int rsx_sort_pairs_inplace_async_ws(void *d_keys, void *d_keys_scratch, void *d_vals, void *d_vals_scratch, size_t n, rsx_dtype dtype,
                                    size_t payload_bytes, rsx_order order, void *d_workspace, size_t workspace_bytes, void *stream)
{
	const size_t kb = dtype_size(dtype);
	if (!kb || (payload_bytes != 4 && payload_bytes != 8) || (n && (!d_keys || !d_keys_scratch || !d_vals || !d_vals_scratch)))
		return fail(RSX_EINVAL, "rsx_sort_pairs_inplace_async_ws: bad argument");
	if (n < 2)
		return RSX_OK;
	Ctx view;
	size_t status_total = 0;
	if (payload_bytes == 4) {
		RSX_DISPATCH_KT(dtype, status_total = (status_bytes<KT, u32>(n) * sizeof(KT)));
	} else {
		RSX_DISPATCH_KT(dtype, status_total = (status_bytes<KT, u64>(n) * sizeof(KT)));
	}
	RSX_TRY(borrow_ctx(view, stream, d_workspace, workspace_bytes, n, kb, status_total));
	if (payload_bytes == 4) {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_inplace_async<KT, u32>(view, (KT *)d_keys, (KT *)d_keys_scratch, (u32 *)d_vals,
		                                                                 (u32 *)d_vals_scratch, n, dtype, order)));
	} else {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_inplace_async<KT, u64>(view, (KT *)d_keys, (KT *)d_keys_scratch, (u64 *)d_vals,
		                                                                 (u64 *)d_vals_scratch, n, dtype, order)));
	}
	return RSX_OK;
}

void rsx_release_stream(void *stream)
{
	// Lock order everywhere else: a context's mutex, then g_mu (the host wrappers hold the context and look contexts up
	// again).  So the context is taken out of the table under g_mu alone, and only then locked, drained and freed.  The
	// caller must not have another thread inside the library on this (device, stream) -- see rsx.h.
	Ctx *c = nullptr;
	{
		std::lock_guard<std::mutex> lock(g_mu);
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) {
			(void)hipGetLastError();
			return;
		}
		auto it = g_ctx.find(std::make_pair(dev, stream));
		if (it == g_ctx.end())
			return;
		c = it->second;
		g_ctx.erase(it);
	}
	{
		std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);   // (a call that was already running on it finishes first)
		(void)hipStreamSynchronize(c->stream);
		c->release();
	}
	delete c;
}

int rsx_sort_pairs_inplace_async(void *d_keys, void *d_keys_scratch, void *d_vals, void *d_vals_scratch, size_t n, rsx_dtype dtype,
                                 size_t payload_bytes, rsx_order order, void *stream)
{
	if (!dtype_size(dtype) || (payload_bytes != 4 && payload_bytes != 8) ||
	    (n && (!d_keys || !d_keys_scratch || !d_vals || !d_vals_scratch)))
		return fail(RSX_EINVAL, "rsx_sort_pairs_inplace_async: bad argument");
	if (n < 2)
		return RSX_OK;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	AsyncScope async_scope((hipStream_t)stream);
	if (payload_bytes == 4) {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_inplace_async<KT, u32>(*c, (KT *)d_keys, (KT *)d_keys_scratch, (u32 *)d_vals,
		                                                                 (u32 *)d_vals_scratch, n, dtype, order)));
	} else {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_inplace_async<KT, u64>(*c, (KT *)d_keys, (KT *)d_keys_scratch, (u64 *)d_vals,
		                                                                 (u64 *)d_vals_scratch, n, dtype, order)));
	}
	return RSX_OK;
}

int rsx_sort_device(void *d_src, void *d_aux, size_t n, rsx_dtype dtype, rsx_order order, void *stream, void **result,
                    rsx_info *info)
{
	info_clear(info, dtype);
	if (!dtype_size(dtype) || !result || (n && (!d_src || !d_aux)))
		return fail(RSX_EINVAL, "rsx_sort_device: bad argument");
	if (n < 2) {                             // radix_sort.hpp:100-101
		*result = d_src;
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	RSX_DISPATCH_KT(dtype, return sort_keys_device<KT>(*c, (KT *)d_src, (KT *)d_aux, n, dtype, order, result, info));
	return RSX_OK;
}

int rsx_sort_pairs_device(void *d_keys, void *d_keys_aux, void *d_vals, void *d_vals_aux, size_t n, rsx_dtype dtype,
                          size_t payload_bytes, rsx_order order, void *stream, rsx_info *info)
{
	info_clear(info, dtype);
	if (!dtype_size(dtype) || (payload_bytes != 4 && payload_bytes != 8) ||
	    (n && (!d_keys || !d_keys_aux || !d_vals || !d_vals_aux)))
		return fail(RSX_EINVAL, "rsx_sort_pairs_device: bad argument");
	if (n < 2) {
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	if (payload_bytes == 4) {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_device<KT, u32>(*c, (KT *)d_keys, (KT *)d_keys_aux, (u32 *)d_vals,
		                                                         (u32 *)d_vals_aux, n, dtype, order, info)));
	} else {
		RSX_DISPATCH_KT(dtype, return (sort_pairs_device<KT, u64>(*c, (KT *)d_keys, (KT *)d_keys_aux, (u64 *)d_vals,
		                                                         (u64 *)d_vals_aux, n, dtype, order, info)));
	}
	return RSX_OK;
}

int rsx_sort_rank_inplace_async(const void *d_src, void *d_index_buffer, size_t n, rsx_dtype dtype, size_t idx_bytes,
                                rsx_order order, void *stream)
{
	if (!dtype_size(dtype) || (idx_bytes != 4 && idx_bytes != 8) || (n && (!d_src || !d_index_buffer)))
		return fail(RSX_EINVAL, "rsx_sort_rank_inplace_async: bad argument");
	if (idx_bytes == 4 && n > (1ull << 32))
		return fail(RSX_EINVAL, "rsx_sort_rank_inplace_async: n does not fit a 4-byte index");
	if (n == 0)
		return RSX_OK;                       // radix_sort_rank.hpp:28-32
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	AsyncScope async_scope((hipStream_t)stream);
	if (n == 1) {
		HIP_TRY(hipMemsetAsync(d_index_buffer, 0, idx_bytes, c->stream));
		return RSX_OK;
	}
	if (idx_bytes == 4) {
		RSX_DISPATCH_KT(dtype, return (sort_rank_inplace_async<KT, u32>(*c, (const KT *)d_src, (u32 *)d_index_buffer, n, dtype, order)));
	} else {
		RSX_DISPATCH_KT(dtype, return (sort_rank_inplace_async<KT, u64>(*c, (const KT *)d_src, (u64 *)d_index_buffer, n, dtype, order)));
	}
	return RSX_OK;
}

int rsx_verify_poll(void *stream, uint64_t *mismatches)
{
	if (mismatches)
		*mismatches = 0;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (!c->vasync.p)
		return RSX_OK;
	u64 bad = 0;
	HIP_TRY(hipMemcpy(&bad, c->vasync.p, sizeof(bad), hipMemcpyDeviceToHost));
	if (mismatches)
		*mismatches = bad;
	if (bad) {
		HIP_TRY(hipMemset(c->vasync.p, 0, 8));
		return fail(RSX_EVERIFY, "RSX_VERIFY: a device-scheduled sort on this stream had a pass whose checked tile differs from its "
		                         "ballot-ranked re-computation in %llu places", (unsigned long long)bad);
	}
	return RSX_OK;
}

int rsx_async_route(void *stream, uint32_t *route)
{
	if (!route)
		return fail(RSX_EINVAL, "rsx_async_route: bad argument");
	*route = 0;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->async_small)
		return RSX_OK;   // (the one-launch sort writes no device-side plan: what lies there is an earlier sort's)
	Plan plan;
	HIP_TRY(hipMemcpy(&plan, c->plan(), sizeof(plan), hipMemcpyDeviceToHost));
	if (c->async_tried_blind && c->seg.p) {
		SegCtl ctl;
		HIP_TRY(hipMemcpy(&ctl, c->seg.p, sizeof(ctl), hipMemcpyDeviceToHost));
		if (getenv("RSX_DEBUG_ROUTE"))   // (what the device left behind: which test ended an attempt)
			fprintf(stderr, "rsx_async_route: blind %u mode %u overflow %u ntiles %u nleaf %u maxleaf %u shift1 %u shift2 %u cmask %08x%08x "
			                "narrow %u compact %u leaf16 %u boff_skip %u boff_next %u slots in the second buffer %u cap1 %u cap2 %u\n",
			        ctl.blind, ctl.mode, ctl.overflow, ctl.ntiles, ctl.nleaf, ctl.maxleaf, ctl.shift1, ctl.shift2, ctl.cmask_hi, ctl.cmask_lo,
			        ctl.narrow, ctl.compact, ctl.leaf16, ctl.boff_skip, ctl.boff_next, c->slack1_lo, c->slack1_cap, c->slack_cap);
		if (ctl.mode == SEG_MODE_LEAVES) {
			*route = 5;
			return RSX_OK;
		}
	}
	if (!plan.sorted && plan.hyb == HYB_ONE_LEVEL)
		*route = 1;
	return RSX_OK;
}

int rsx_sort_rank_device(const void *d_src, void *d_index_buffer, size_t n, rsx_dtype dtype, size_t idx_bytes,
                         rsx_order order, void *stream, void **result, rsx_info *info)
{
	info_clear(info, dtype);
	if (!dtype_size(dtype) || (idx_bytes != 4 && idx_bytes != 8) || !result || (n && (!d_src || !d_index_buffer)))
		return fail(RSX_EINVAL, "rsx_sort_rank_device: bad argument");
	if (idx_bytes == 4 && n > (1ull << 32))
		return fail(RSX_EINVAL, "rsx_sort_rank_device: n does not fit a 4-byte index");
	*result = d_index_buffer;
	if (n == 0) {                            // radix_sort_rank.hpp:28-32
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	if (n == 1) {
		HIP_TRY(hipMemsetAsync(d_index_buffer, 0, idx_bytes, c->stream));
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	if (idx_bytes == 4) {
		RSX_DISPATCH_KT(dtype, return (sort_rank_device<KT, u32>(*c, (const KT *)d_src, (u32 *)d_index_buffer, n, dtype,
		                                                        order, result, info)));
	} else {
		RSX_DISPATCH_KT(dtype, return (sort_rank_device<KT, u64>(*c, (const KT *)d_src, (u64 *)d_index_buffer, n, dtype,
		                                                        order, result, info)));
	}
	return RSX_OK;
}

// host arrays the one-launch kernels take (sort_keys_device, sort_rank_device: the same conditions)
static bool host_small_path(const Ctx &c, size_t key_bytes)
{
	return c.fast && key_bytes <= SMALL_SORT_BYTES && !env().no_small_sort && !env().no_host_small && !capture_armed();
}

int rsx_sort(void *src, void *aux, size_t n, rsx_dtype dtype, rsx_order order, void **result, rsx_info *info)
{
	info_clear(info, dtype);
	const size_t kb = dtype_size(dtype);
	if (!kb || !result || (n && (!src || !aux)))
		return fail(RSX_EINVAL, "rsx_sort: bad argument");
	if (n < 2) {                             // radix_sort.hpp:100-101
		*result = src;
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(nullptr, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	if (is_device_ptr(src)) {
		if (!is_device_ptr(aux))
			return fail(RSX_EINVAL, "rsx_sort: src is a device pointer but aux is not");
		RSX_TRY(rsx_sort_device(src, aux, n, dtype, order, nullptr, result, info));
		HIP_TRY(hipStreamSynchronize(c->stream));
		return RSX_OK;
	}
	if (host_small_path(*c, n * kb)) {
		// small host arrays: the one-launch kernel reads the keys from pinned memory and writes the result there
		RSX_TRY(c->ensure_hstage());
		memcpy(c->hstage, src, n * kb);
		void *dres = nullptr;
		rsx_info li;
		RSX_TRY(rsx_sort_device(c->hstage_dev, c->hstage_dev + SMALL_SORT_BYTES, n, dtype, order, nullptr, &dres, &li));
		if (info)
			*info = li;
		if (li.early_exit) {
			*result = src;
			return RSX_OK;
		}
		void *hres = li.result_in_aux ? aux : src;
		memcpy(hres, c->hstage + ((char *)dres - c->hstage_dev), n * kb);
		*result = hres;
		return RSX_OK;
	}
	// host buffers: stage over PCIe, sort in HBM, bring the result back into the
	// buffer the returned-pointer rule names (the other one is left as it was)
	RSX_TRY(c->keys[0].ensure(n * kb));
	RSX_TRY(c->keys[1].ensure(n * kb));
	HostRegScope reg_src(src, n * kb), reg_aux(aux, n * kb);   // (RSX_HOST_REGISTER=1: pinned until this call returns)
	HIP_TRY(hipMemcpyAsync(c->keys[0].p, src, n * kb, hipMemcpyHostToDevice, c->stream));
	void *dres = nullptr;
	rsx_info li;
	RSX_TRY(rsx_sort_device(c->keys[0].p, c->keys[1].p, n, dtype, order, nullptr, &dres, &li));
	if (info)
		*info = li;
	if (li.early_exit) {
		HIP_TRY(hipStreamSynchronize(c->stream));
		*result = src;
		return RSX_OK;
	}
	void *hres = li.result_in_aux ? aux : src;
	HIP_TRY(hipMemcpyAsync(hres, dres, n * kb, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	*result = hres;
	return RSX_OK;
}

int rsx_sort_rank(const void *src, void *index_buffer, size_t n, rsx_dtype dtype, size_t idx_bytes, rsx_order order,
                  void **result, rsx_info *info)
{
	info_clear(info, dtype);
	const size_t kb = dtype_size(dtype);
	if (!kb || !result || (idx_bytes != 1 && idx_bytes != 2 && idx_bytes != 4 && idx_bytes != 8) ||
	    (n && (!src || !index_buffer)))
		return fail(RSX_EINVAL, "rsx_sort_rank: bad argument");
	if (idx_bytes < 8 && n > (1ull << (8 * idx_bytes)))
		return fail(RSX_EINVAL, "rsx_sort_rank: n = %zu does not fit a %zu-byte index", n, idx_bytes);
	*result = index_buffer;
	if (n == 0) {
		if (info)
			info->early_exit = 1;
		return RSX_OK;
	}
	Ctx *c;
	RSX_TRY(get_ctx(nullptr, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	const bool dev = is_device_ptr(src);
	if (dev != is_device_ptr(index_buffer))
		return fail(RSX_EINVAL, "rsx_sort_rank: src and index_buffer must both be host or both be device pointers");
	if (dev && idx_bytes >= 4) {
		RSX_TRY(rsx_sort_rank_device(src, index_buffer, n, dtype, idx_bytes, order, nullptr, result, info));
		HIP_TRY(hipStreamSynchronize(c->stream));
		return RSX_OK;
	}
	const size_t wide = idx_bytes == 8 ? 8 : 4;
	if (!dev && host_small_path(*c, 0) && n * 2 * (kb + wide) <= SMALL_PAIR_BYTES) {
		// small host arrays: keys read from and ranks written to pinned memory by the one-launch kernel, narrowed here
		RSX_TRY(c->ensure_hstage());
		memcpy(c->hstage, src, n * kb);
		char *hib = c->hstage + SMALL_SORT_BYTES, *dib = c->hstage_dev + SMALL_SORT_BYTES;
		void *dres = nullptr;
		rsx_info li;
		RSX_TRY(rsx_sort_rank_device(c->hstage_dev, dib, n, dtype, wide, order, nullptr, &dres, &li));
		if (info)
			*info = li;
		const bool second = dres != (void *)dib;
		char *out = (char *)index_buffer + (second ? n * idx_bytes : 0);
		const char *from = hib + ((char *)dres - dib);
		if (idx_bytes >= 4) {
			memcpy(out, from, n * idx_bytes);
		} else if (idx_bytes == 2) {
			for (size_t i = 0; i < n; ++i)
				((uint16_t *)out)[i] = (uint16_t)((const u32 *)from)[i];
		} else {
			for (size_t i = 0; i < n; ++i)
				((uint8_t *)out)[i] = (uint8_t)((const u32 *)from)[i];
		}
		*result = out;
		return RSX_OK;
	}
	// staged path: keys in recs[0] (host keys) or in place (device keys); indices computed
	// as 4- or 8-byte values in vals[0], narrowed into vals[1] when IdxType is 1 or 2 bytes
	const void *dkeys = src;
	if (!dev) {
		RSX_TRY(c->recs[0].ensure(n * kb));
		HIP_TRY(hipMemcpyAsync(c->recs[0].p, src, n * kb, hipMemcpyHostToDevice, c->stream));
		dkeys = c->recs[0].p;
	}
	RSX_TRY(c->vals[0].ensure(2 * n * wide));
	void *dres = nullptr;
	rsx_info li;
	RSX_TRY(rsx_sort_rank_device(dkeys, c->vals[0].p, n, dtype, wide, order, nullptr, &dres, &li));
	if (info)
		*info = li;
	const bool second = dres != c->vals[0].p;
	char *out = (char *)index_buffer + (second ? n * idx_bytes : 0);
	const void *from = dres;
	if (idx_bytes < 4) {
		RSX_TRY(c->vals[1].ensure(n * idx_bytes));
		if (idx_bytes == 1)
			hipLaunchKernelGGL((rsx_convert_kernel<uint8_t, u32>), dim3(256), dim3(256), 0, c->stream, (uint8_t *)c->vals[1].p,
			                   (const u32 *)dres, (u64)n);
		else
			hipLaunchKernelGGL((rsx_convert_kernel<uint16_t, u32>), dim3(256), dim3(256), 0, c->stream,
			                   (uint16_t *)c->vals[1].p, (const u32 *)dres, (u64)n);
		HIP_TRY(hipGetLastError());
		from = c->vals[1].p;
	}
	HIP_TRY(hipMemcpyAsync(out, from, n * idx_bytes, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	*result = out;
	return RSX_OK;
}

#include "rsx_records.hpp"       // rsx_sort_rank_keys, rsx_sort_records, rsx_sort_records_tagged[_device]

#include "rsx_multi_entry.hpp"   // rsx_sort_multi

int rsx_histogram_device(const void *d_src, size_t n, rsx_dtype dtype, rsx_order order, uint64_t *d_hist,
                         uint32_t *d_unsorted, void *stream)
{
	const size_t kb = dtype_size(dtype);
	if (!kb || !d_hist || !d_unsorted || (n && !d_src))
		return fail(RSX_EINVAL, "rsx_histogram_device: bad argument");
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * kb * sizeof(u64), c->stream));
	HIP_TRY(hipMemsetAsync(d_unsorted, 0, sizeof(u32), c->stream));
	if (n == 0)
		return RSX_OK;
	RSX_DISPATCH_KT(dtype, return launch_hist<KT>(*c, (const KT *)d_src, n, make_kdf<KT>(dtype, order), (u64 *)d_hist,
	                                              (u32 *)d_unsorted));
	return RSX_OK;
}

// The MSD split of the multi-GPU sort as ONE ordinary scatter pass on the top KDF byte.  The destinations of the exchange are
// contiguous ranges of that byte (multi.py, choose_splitters), so a stable pass by the byte itself leaves every
// destination's keys contiguous in d_dst -- with 256 digits instead of G buckets there is no crowd of lanes on a handful
// of LDS counters, and the pass runs on the plain-digit kernel.  Keys of one destination arrive ordered by (top byte,
// original index) instead of by original index; equal keys have equal top bytes, so the stable order of the final result
// is the same.  top_hist (host, 256 uint64) receives the counts of the byte; the pass itself is only enqueued.  `column`
// picks another byte than the top one (a caller that knows the top bytes to be constant splits by the highest varying one).
int rsx_msd_split_device(const void *d_src, void *d_dst, size_t n, rsx_dtype dtype, rsx_order order, int column,
                         uint64_t *top_hist, void *stream)
{
	const size_t kb = dtype_size(dtype);
	if (!kb || !top_hist || column >= (int)kb || (n && (!d_src || !d_dst)))
		return fail(RSX_EINVAL, "rsx_msd_split_device: bad argument");
	const u32 col = column < 0 ? (u32)kb - 1 : (u32)column;
	for (int i = 0; i < 256; ++i)
		top_hist[i] = 0;
	if (n == 0)
		return RSX_OK;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	const size_t hist_bytes = kb * 256 * sizeof(u64);
	HIP_TRY(hipMemsetAsync(c->ghist(), 0, hist_bytes, c->stream));
	HIP_TRY(hipMemsetAsync(c->small_set(), 0, 256, c->stream));
	RSX_DISPATCH_KT(dtype, return msd_split<KT>(*c, (const KT *)d_src, (KT *)d_dst, n, dtype, order, col, top_hist));
	return RSX_OK;
}

int rsx_msd_split_async(const void *d_src, void *d_dst, size_t n, rsx_dtype dtype, rsx_order order, int column,
                        const uint64_t *d_hist, void *stream)
{
	const size_t kb = dtype_size(dtype);
	const bool hot = column >= 0 && (column & RSX_SPLIT_HOT) != 0;
	if (column >= 0)
		column &= ~RSX_SPLIT_HOT;
	if (!kb || !d_hist || column >= (int)kb || (n && (!d_src || !d_dst)))
		return fail(RSX_EINVAL, "rsx_msd_split_async: bad argument");
	const u32 col = column < 0 ? (u32)kb - 1 : (u32)column;
	if (n == 0)
		return RSX_OK;
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	AsyncScope async_scope((hipStream_t)stream);
	HIP_TRY(hipMemsetAsync(c->small_set(), 0, 256, c->stream));
	RSX_DISPATCH_KT(dtype, return msd_split_known<KT>(*c, (const KT *)d_src, (KT *)d_dst, n, dtype, order, col, (const u64 *)d_hist, hot));
	return RSX_OK;
}

void rsx_reload_env(void)
{
	std::lock_guard<std::mutex> lock(g_mu);
	(void)env();
	g_env.load();
	g_env_epoch.fetch_add(1u);   // (contexts forget what they have learnt about their inputs: sort_keys_blind's back-off)
}

int rsx_profile_begin(void)
{
	std::lock_guard<std::mutex> lock(g_mu);
	for (auto &r : g_prof) {
		(void)hipEventDestroy(r.start);
		(void)hipEventDestroy(r.stop);
	}
	g_prof.clear();
	g_prof_on = true;
	return RSX_OK;
}

int rsx_profile_end(rsx_profile *out)
{
	std::lock_guard<std::mutex> lock(g_mu);
	g_prof_on = false;
	if (!out)
		return fail(RSX_EINVAL, "rsx_profile_end: null output");
	memset(out, 0, sizeof(*out));
	for (auto &r : g_prof) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(r.stop));
		HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
		if (r.verdict && ((r.valid_if == 1) != (*r.verdict == SEG_MODE_LEAVES)))
			r.called_off = true;
		if (r.called_off) {
			out->called_off_ms += ms;
			out->called_off_launches += 1;
		} else if (r.kind == 0) {
			out->hist_ms += ms;
			out->hist_launches += 1;
			out->hist_bytes += r.bytes;
		} else if (r.kind == 3) {
			out->narrow_ms += ms;
			out->narrow_launches += 1;
			out->narrow_bytes += r.bytes;
		} else if (r.kind == 2) {
			out->leaf_ms += ms;
			out->leaf_launches += 1;
			out->leaf_bytes += r.bytes;
		} else {
			out->scatter_ms += ms;
			out->scatter_launches += 1;
			out->scatter_bytes += r.bytes;
		}
		(void)hipEventDestroy(r.start);
		(void)hipEventDestroy(r.stop);
	}
	g_prof.clear();
	g_prof_vnext = 0;
	return RSX_OK;
}

int rsx_fill_splitmix_device(void *d_dst, size_t n, size_t elem_bytes, uint64_t seed, uint64_t mask, uint64_t first_index,
                             void *stream)
{
	if (n && !d_dst)
		return fail(RSX_EINVAL, "rsx_fill_splitmix_device: null destination");
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	std::lock_guard<std::recursive_mutex> ctx_lock(c->mu);
	if (n == 0)
		return RSX_OK;
	const dim3 grid(2048), block(256);
	switch (elem_bytes) {
	case 1: hipLaunchKernelGGL((rsx_fill_splitmix_kernel<uint8_t>), grid, block, 0, c->stream, (uint8_t *)d_dst, (u64)n, (u64)seed, (u64)mask, (u64)first_index); break;
	case 2: hipLaunchKernelGGL((rsx_fill_splitmix_kernel<uint16_t>), grid, block, 0, c->stream, (uint16_t *)d_dst, (u64)n, (u64)seed, (u64)mask, (u64)first_index); break;
	case 4: hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u32>), grid, block, 0, c->stream, (u32 *)d_dst, (u64)n, (u64)seed, (u64)mask, (u64)first_index); break;
	case 8: hipLaunchKernelGGL((rsx_fill_splitmix_kernel<u64>), grid, block, 0, c->stream, (u64 *)d_dst, (u64)n, (u64)seed, (u64)mask, (u64)first_index); break;
	default: return fail(RSX_EINVAL, "rsx_fill_splitmix_device: elem_bytes must be 1, 2, 4 or 8");
	}
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

int rsx_spin_device(uint64_t microseconds, void *stream)
{
	if (microseconds > 1000000)
		return fail(RSX_EINVAL, "rsx_spin_device: at most one second");
	Ctx *c;
	RSX_TRY(get_ctx(stream, &c));
	hipLaunchKernelGGL(rsx_spin_kernel, dim3(1), dim3(64), 0, c->stream, (u64)microseconds * 100);
	HIP_TRY(hipGetLastError());
	return RSX_OK;
}

}  // extern "C"
