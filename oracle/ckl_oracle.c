/* TEST INFRASTRUCTURE ONLY — see ckl_oracle.h.
 *
 * Plain-C restatement of the reference CPU algorithm for
 *   crackle::compress   (src/crackle.hpp:34-257)
 *   crackle::decompress (src/crackle.hpp:503-663)
 * Every function cites the reference file:line it follows.  Nothing here is
 * linked into, imported by or executed from the product path (crackle_amd/).
 *
 * Parity status: PINNED against oracle/_ref (the reference compiled in place)
 * and the committed fixtures under tests/golden/.
 */
#define _GNU_SOURCE
#include "ckl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static _Thread_local char g_err[512];

#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return 1; } while (0)

const char* ckl_oracle_last_error(void) { return g_err; }
void ckl_oracle_free(void* p) { free(p); }

static void* xmalloc(size_t n) {
	void* p = malloc(n ? n : 1);
	if (!p) { fprintf(stderr, "ckl_oracle: out of memory (%zu)\n", n); abort(); }
	return p;
}
static void* xcalloc(size_t n, size_t sz) {
	void* p = calloc(n ? n : 1, sz ? sz : 1);
	if (!p) { fprintf(stderr, "ckl_oracle: out of memory\n"); abort(); }
	return p;
}
static void* xrealloc(void* q, size_t n) {
	void* p = realloc(q, n ? n : 1);
	if (!p) { fprintf(stderr, "ckl_oracle: out of memory\n"); abort(); }
	return p;
}

/* ------------------------------------------------------------------ */
/* byte buffers and little-endian helpers (src/lib.hpp:11-145)          */
/* ------------------------------------------------------------------ */
typedef struct { unsigned char* p; size_t n, cap; } bytes_t;

static void breserve(bytes_t* b, size_t extra) {
	if (b->n + extra > b->cap) {
		size_t c = b->cap ? b->cap * 2 : 64;
		while (c < b->n + extra) c *= 2;
		b->p = (unsigned char*)xrealloc(b->p, c);
		b->cap = c;
	}
}
static void bpush(bytes_t* b, const void* src, size_t n) {
	breserve(b, n);
	if (n) memcpy(b->p + b->n, src, n);
	b->n += n;
}
static void bput8(bytes_t* b, uint8_t v) { breserve(b, 1); b->p[b->n++] = v; }
/* itocd (src/lib.hpp:11-18): LE, dynamic width */
static void bput(bytes_t* b, uint64_t v, int width) {
	breserve(b, (size_t)width);
	for (int i = 0; i < width; i++) b->p[b->n++] = (unsigned char)((v >> (8 * i)) & 0xFF);
}
/* ctoid (src/lib.hpp:137-145) restated without the int-promotion bug */
static uint64_t rd(const unsigned char* buf, uint64_t idx, int width) {
	uint64_t v = 0;
	for (int i = 0; i < width; i++) v |= ((uint64_t)buf[idx + i]) << (8 * i);
	return v;
}
/* compute_byte_width (src/lib.hpp:236-247) */
static int byte_width(uint64_t x) {
	if (x <= 0xFFull) return 1;
	if (x <= 0xFFFFull) return 2;
	if (x <= 0xFFFFFFFFull) return 4;
	return 8;
}
static int ilog2(int w) { return w == 1 ? 0 : w == 2 ? 1 : w == 4 ? 2 : 3; }

/* ------------------------------------------------------------------ */
/* CRCs (src/crc.hpp:23-57; third_party/fastcrc: CRC-32C Castagnoli)   */
/* ------------------------------------------------------------------ */
static uint8_t crc8(const uint8_t* data, uint64_t size) {
	const uint8_t polynomial = 0xe7;
	uint8_t crc = 0xFF;
	while (size--) {
		crc ^= *data++;
		for (int k = 0; k < 8; k++) crc = (crc & 1) ? (uint8_t)((crc >> 1) ^ polynomial) : (uint8_t)(crc >> 1);
	}
	return crc;
}

static uint32_t g_crc_tab[8][256];
static pthread_once_t g_crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void) {
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
		g_crc_tab[0][i] = c;
	}
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = g_crc_tab[0][i];
		for (int t = 1; t < 8; t++) {
			c = g_crc_tab[0][c & 0xFF] ^ (c >> 8);
			g_crc_tab[t][i] = c;
		}
	}
}
/* crc32_impl(0, data, n): init ~0, xorout ~0, reflected 0x82F63B78 */
uint32_t ckl_oracle_crc32c(const uint8_t* data, uint64_t n) {
	pthread_once(&g_crc_once, crc_init);
	uint32_t crc = 0xFFFFFFFFu;
	while (n && ((uintptr_t)data & 7)) {
		crc = g_crc_tab[0][(crc ^ *data++) & 0xFF] ^ (crc >> 8);
		n--;
	}
	while (n >= 8) {
		uint64_t w;
		memcpy(&w, data, 8);
		w ^= crc;
		crc = g_crc_tab[7][w & 0xFF] ^ g_crc_tab[6][(w >> 8) & 0xFF]
			^ g_crc_tab[5][(w >> 16) & 0xFF] ^ g_crc_tab[4][(w >> 24) & 0xFF]
			^ g_crc_tab[3][(w >> 32) & 0xFF] ^ g_crc_tab[2][(w >> 40) & 0xFF]
			^ g_crc_tab[1][(w >> 48) & 0xFF] ^ g_crc_tab[0][(w >> 56) & 0xFF];
		data += 8;
		n -= 8;
	}
	while (n--) crc = g_crc_tab[0][(crc ^ *data++) & 0xFF] ^ (crc >> 8);
	return ~crc;
}

/* ------------------------------------------------------------------ */
/* parallel-for over z (src/threadpool.hpp:49-133: one task per slice)  */
/* ------------------------------------------------------------------ */
typedef void (*slice_fn)(int64_t z, size_t tid, void* ctx);
typedef struct { slice_fn fn; void* ctx; atomic_llong next; int64_t n; } pf_t;
typedef struct { pf_t* pf; size_t tid; } pf_arg_t;

static void* pf_worker(void* a) {
	pf_arg_t* arg = (pf_arg_t*)a;
	pf_t* pf = arg->pf;
	for (;;) {
		long long z = atomic_fetch_add(&pf->next, 1);
		if (z >= pf->n) break;
		pf->fn(z, arg->tid, pf->ctx);
	}
	return NULL;
}
static void parallel_for(int64_t n, size_t threads, slice_fn fn, void* ctx) {
	if (threads <= 1 || n <= 1) {
		for (int64_t z = 0; z < n; z++) fn(z, 0, ctx);
		return;
	}
	pf_t pf = { fn, ctx, 0, n };
	pthread_t* th = (pthread_t*)xmalloc(sizeof(pthread_t) * threads);
	pf_arg_t* args = (pf_arg_t*)xmalloc(sizeof(pf_arg_t) * threads);
	for (size_t t = 0; t < threads; t++) {
		args[t].pf = &pf;
		args[t].tid = t;
		pthread_create(&th[t], NULL, pf_worker, &args[t]);
	}
	for (size_t t = 0; t < threads; t++) pthread_join(th[t], NULL);
	free(th);
	free(args);
}
static size_t resolve_parallel(uint64_t parallel, int64_t sz) {
	/* src/crackle.hpp:66-69, 570-573 */
	if (parallel == 0) {
		long n = sysconf(_SC_NPROCESSORS_ONLN);
		parallel = n > 0 ? (uint64_t)n : 1;
	}
	if ((int64_t)parallel > sz) parallel = (uint64_t)sz;
	if (parallel == 0) parallel = 1;
	return (size_t)parallel;
}

/* ------------------------------------------------------------------ */
/* header (src/header.hpp:35-308)                                      */
/* ------------------------------------------------------------------ */
enum { FLAT = 0, PINS_FIXED_WIDTH = 1, PINS_VARIABLE_WIDTH = 2 };
enum { IMPERMISSIBLE = 0, PERMISSIBLE = 1 };

typedef struct {
	uint8_t format_version;
	int label_format, crack_format, is_signed;
	int data_width, stored_data_width;
	uint32_t sx, sy, sz;
	uint8_t log2_grid_size;
	uint64_t num_label_bytes;
	int fortran_order, markov_model_order, is_sorted;
	uint8_t crc;
} header_t;

#define HEADER_BYTES 29
#define HEADER_BYTES_V0 24

/* assign_from_buffer (src/header.hpp:98-150) */
static int header_read(header_t* h, const unsigned char* buf, uint64_t n) {
	if (n < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream. Bytes: %llu", (unsigned long long)n);
	int valid_magic = (buf[0] == 'c' && buf[1] == 'r' && buf[2] == 'k' && buf[3] == 'l');
	h->format_version = buf[4];
	if (!valid_magic || h->format_version > 1) FAIL("crackle: Data stream is not valid. Unable to decompress.");
	if (h->format_version == 1 && n < HEADER_BYTES) FAIL("crackle: Input too small to be a valid stream. Bytes: %llu", (unsigned long long)n);
	uint16_t fmt = (uint16_t)rd(buf, 5, 2);
	h->sx = (uint32_t)rd(buf, 7, 4);
	h->sy = (uint32_t)rd(buf, 11, 4);
	h->sz = (uint32_t)rd(buf, 15, 4);
	h->log2_grid_size = buf[19];
	h->num_label_bytes = h->format_version == 0 ? rd(buf, 20, 4) : rd(buf, 20, 8);
	h->data_width = 1 << (fmt & 3);
	h->stored_data_width = 1 << ((fmt >> 2) & 3);
	h->crack_format = (fmt >> 4) & 1;
	h->label_format = (fmt >> 5) & 3;
	h->fortran_order = (fmt >> 7) & 1;
	h->is_signed = (fmt >> 8) & 1;
	h->markov_model_order = (fmt >> 9) & 15;
	h->is_sorted = !((fmt >> 13) & 1);
	if (h->format_version == 0) return 0;
	h->crc = buf[28];
	if (crc8(buf + 5, 28 - 5) != h->crc) FAIL("crackle: CRC8 check failed. Header may be corrupted.");
	return 0;
}
/* tochars (src/header.hpp:206-267), always v1 */
static void header_write(const header_t* h, bytes_t* out) {
	size_t base = out->n;
	bpush(out, "crkl", 4);
	uint16_t fmt = 0;
	fmt |= (uint16_t)ilog2(h->data_width);
	fmt |= (uint16_t)(ilog2(h->stored_data_width) << 2);
	fmt |= (uint16_t)(h->crack_format << 4);
	fmt |= (uint16_t)(h->label_format << 5);
	fmt |= (uint16_t)((h->fortran_order ? 1 : 0) << 7);
	fmt |= (uint16_t)((h->is_signed ? 1 : 0) << 8);
	fmt |= (uint16_t)((h->markov_model_order & 15) << 9);
	fmt |= (uint16_t)((h->is_sorted ? 0 : 1) << 13);
	bput8(out, 1);
	bput(out, fmt, 2);
	bput(out, h->sx, 4);
	bput(out, h->sy, 4);
	bput(out, h->sz, 4);
	bput8(out, h->log2_grid_size);
	bput(out, h->num_label_bytes, 8);
	bput8(out, crc8(out->p + base + 5, HEADER_BYTES - 1 - 5));
}
static uint64_t header_bytes(const header_t* h) { return h->format_version == 0 ? HEADER_BYTES_V0 : HEADER_BYTES; }
static uint64_t grid_index_bytes(const header_t* h) {
	return h->format_version == 0 ? (uint64_t)h->sz * 4 : ((uint64_t)h->sz + 1) * 4;
}
/* src/header.hpp:284-297 */
static uint64_t markov_model_bytes(const header_t* h) {
	if (h->markov_model_order == 0) return 0;
	uint64_t model_size = 1ull << (2 * h->markov_model_order);
	return (model_size * 5 + 4) / 8;
}
/* src/header.hpp:190-192: NOTE 32-bit product */
static int pin_index_width(const header_t* h) {
	uint32_t v = h->sx * h->sy * h->sz;
	return byte_width(v);
}

/* ------------------------------------------------------------------ */
/* labels widened to u64 (the reference is templated on LABEL)         */
/* ------------------------------------------------------------------ */
static uint64_t* widen(const void* labels, int w, uint64_t n) {
	uint64_t* out = (uint64_t*)xmalloc(sizeof(uint64_t) * n);
	if (w == 1) { const uint8_t* p = labels; for (uint64_t i = 0; i < n; i++) out[i] = p[i]; }
	else if (w == 2) { const uint16_t* p = labels; for (uint64_t i = 0; i < n; i++) out[i] = p[i]; }
	else if (w == 4) { const uint32_t* p = labels; for (uint64_t i = 0; i < n; i++) out[i] = p[i]; }
	else { memcpy(out, labels, 8 * n); }
	return out;
}

/* ------------------------------------------------------------------ */
/* 2-D 4-connected CCL (src/cc3d.hpp:42-144, 257-369)                  */
/* dense ids in first-raster-pixel order          */
/* ------------------------------------------------------------------ */
static uint32_t uf_root(uint32_t* ids, uint32_t n) {
	/* DisjointSet::root (src/cc3d.hpp:62-70), path halving */
	uint32_t i = ids[n];
	while (i != ids[i]) {
		ids[i] = ids[ids[i]];
		i = ids[i];
	}
	return i;
}
static void uf_unify(uint32_t* ids, uint32_t p, uint32_t q) {
	/* DisjointSet::unify (src/cc3d.hpp:87-106) */
	if (p == q) return;
	uint32_t i = uf_root(ids, p), j = uf_root(ids, q);
	ids[i] = j;
}
/* relabel (src/cc3d.hpp:114-144): final ids assigned in provisional order */
static uint64_t ccl_relabel(uint32_t* out, int64_t voxels, uint32_t num_labels, uint32_t* ids, uint32_t* renumber, uint64_t start_label) {
	uint32_t next_label = (uint32_t)start_label + 1;
	for (uint32_t i = 0; i <= num_labels; i++) renumber[i] = 0;
	for (uint32_t i = 1; i <= num_labels; i++) {
		uint32_t label = uf_root(ids, i);
		if (renumber[label] == 0) {
			renumber[label] = next_label;
			renumber[i] = next_label;
			next_label++;
		}
		else {
			renumber[i] = renumber[label];
		}
	}
	for (int64_t loc = 0; loc < voxels; loc++) out[loc] = renumber[out[loc]] - 1;
	return next_label - start_label - 1;
}
/* connected_components2d_4 (src/cc3d.hpp:257-369), one slice, on labels.
 * The decision tree of the reference is an optimisation of "unify with the
 * left and the upper neighbour when equal"; provisional labels are created in
 * raster order exactly when a pixel has neither, so numbering is identical. */
static uint64_t ccl_labels_slice(const uint64_t* in, int64_t sx, int64_t sy, uint32_t* out, uint64_t start_label, uint32_t* ids, uint32_t* renumber) {
	uint32_t next = 0;
	for (int64_t y = 0; y < sy; y++) {
		for (int64_t x = 0; x < sx; x++) {
			int64_t loc = x + sx * y;
			int left = (x > 0 && in[loc] == in[loc - 1]);
			int up = (y > 0 && in[loc] == in[loc - sx]);
			if (left) {
				out[loc] = out[loc - 1];
				if (up) uf_unify(ids, out[loc], out[loc - sx]);
			}
			else if (up) {
				out[loc] = out[loc - sx];
			}
			else {
				next++;
				out[loc] = next;
				ids[next] = next;
			}
		}
	}
	return ccl_relabel(out, sx * sy, next, ids, renumber, start_label);
}
/* color_connectivity_graph (src/cc3d.hpp:146-254), one slice, on the VCG:
 * left-connected iff bit1 of the pixel, up-connected iff bit3. */
static uint64_t ccl_vcg_slice(const uint8_t* vcg, int64_t sx, int64_t sy, uint32_t* out, uint32_t* ids, uint32_t* renumber) {
	uint32_t next = 0;
	for (int64_t y = 0; y < sy; y++) {
		for (int64_t x = 0; x < sx; x++) {
			int64_t loc = x + sx * y;
			int left = (x > 0 && (vcg[loc] & 0x2));
			int up = (y > 0 && (vcg[loc] & 0x8));
			if (left) {
				out[loc] = out[loc - 1];
				if (up) uf_unify(ids, out[loc], out[loc - sx]);
			}
			else if (up) {
				out[loc] = out[loc - sx];
			}
			else {
				next++;
				out[loc] = next;
				ids[next] = next;
			}
		}
	}
	return ccl_relabel(out, sx * sy, next, ids, renumber, 0);
}

int ckl_oracle_connected_components(
	const void* labels, int dtype_bytes, int64_t sx, int64_t sy, int64_t sz,
	uint32_t* cc_out, uint64_t* per_slice, uint64_t* N
) {
	/* src/cc3d.hpp:371-400: globally increasing ids */
	const int64_t sxy = sx * sy;
	uint64_t* lab = widen(labels, dtype_bytes, (uint64_t)(sxy * sz));
	uint32_t* ids = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(sxy + 2));
	uint32_t* ren = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(sxy + 2));
	uint64_t total = 0;
	for (int64_t z = 0; z < sz; z++) {
		uint64_t n = ccl_labels_slice(lab + sxy * z, sx, sy, cc_out + sxy * z, total, ids, ren);
		per_slice[z] = n;
		total += n;
	}
	*N = total;
	free(ids); free(ren); free(lab);
	return 0;
}

/* ------------------------------------------------------------------ */
/* crack codes: encode (src/crackcodes.hpp:35-281, 374-496)             */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t node; uint8_t* codes; size_t n; } chain_t;
typedef struct { chain_t* chains; size_t n, cap; } chainset_t;

static void chainset_free(chainset_t* cs) {
	for (size_t i = 0; i < cs->n; i++) free(cs->chains[i].codes);
	free(cs->chains);
	cs->chains = NULL; cs->n = cs->cap = 0;
}
static int chain_cmp(const void* a, const void* b) {
	uint64_t x = ((const chain_t*)a)->node, y = ((const chain_t*)b)->node;
	return x < y ? -1 : x > y;
}

/* remove_initial_branch (src/crackcodes.hpp:185-242) */
static int64_t remove_initial_branch(int64_t node, unsigned char* code, size_t n, int64_t sx) {
	if (n == 0 || code[0] != 'b') return node;
	int64_t i = 1;
	while (code[i] != 't') {
		if (code[i] == 'b') return node;
		i++;
	}
	const int64_t sxe = sx + 1;
	int64_t y = node / sxe, x = node - sxe * y;
	code[0] = 's';
	for (i = 1; code[i] != 't'; i++) {
		switch (code[i]) {
			case 'u': y -= 1; code[i] = 'd'; break;
			case 'd': y += 1; code[i] = 'u'; break;
			case 'l': x -= 1; code[i] = 'r'; break;
			case 'r': x += 1; code[i] = 'l'; break;
			default: break;
		}
	}
	code[i] = 's';
	const int64_t last = i - 1;
	/* reverse code[1..last] */
	for (int64_t a = 1, b = last; a < b; a++, b--) {
		unsigned char t = code[a]; code[a] = code[b]; code[b] = t;
	}
	return x + sxe * y;
}
/* remove_spurious_branches (src/crackcodes.hpp:250-281) */
static void remove_spurious_branches(unsigned char* code, size_t n) {
	int64_t* stack = (int64_t*)xmalloc(sizeof(int64_t) * (n + 2));
	uint32_t* lens = (uint32_t*)xcalloc(n + 2, sizeof(uint32_t));
	int64_t* erase = (int64_t*)xmalloc(sizeof(int64_t) * 2 * (n + 2));
	size_t sp = 0, ne = 0;
	stack[sp++] = -1;
	int64_t current = -1;
	for (int64_t i = 0; i < (int64_t)n; i++) {
		if (code[i] == 'b') {
			stack[sp++] = i;
		}
		else if (code[i] == 't') {
			if (current >= 0 && lens[current + 1] == 0) {
				erase[ne++] = current;
				erase[ne++] = i;
			}
			current = sp ? stack[--sp] : -1;
		}
		else {
			lens[current + 1]++;
		}
	}
	for (size_t k = 0; k < ne; k++) code[erase[k]] = 's';
	free(stack); free(lens); free(erase);
}
/* symbols_to_codepoints (src/crackcodes.hpp:128-183) for one chain */
static size_t symbols_to_codepoints(const unsigned char* sym, size_t n, uint8_t* out) {
	enum { UP = 0, RIGHT = 1, DOWN = 2, LEFT = 3 };
	size_t m = 0;
	for (size_t i = 0; i < n; i++) {
		unsigned char s = sym[i];
		if (s == 's') continue;
		if (s == 'b') {
			if (i > 0 && m > 0 && out[m - 1] != DOWN) { out[m++] = UP; out[m++] = DOWN; }
			else { out[m++] = LEFT; out[m++] = RIGHT; }
		}
		else if (s == 't') {
			if (i > 0 && m > 0 && out[m - 1] != UP) { out[m++] = DOWN; out[m++] = UP; }
			else { out[m++] = RIGHT; out[m++] = LEFT; }
		}
		else if (s == 'u') out[m++] = UP;
		else if (s == 'd') out[m++] = DOWN;
		else if (s == 'l') out[m++] = LEFT;
		else out[m++] = RIGHT;
	}
	return m;
}

/* Graph::init + create_crack_codes (src/crackcodes.hpp:66-125, 374-453) */
static void create_crack_codes(const uint64_t* labels, int64_t sx, int64_t sy, int permissible, chainset_t* cs) {
	const int64_t sxe = sx + 1, sye = sy + 1;
	uint8_t* adj = (uint8_t*)xcalloc((size_t)(sxe * sye), 1);
	int any = 0;
	for (int64_t y = 0; y < sy; y++) {
		for (int64_t x = 0; x < sx; x++) {
			uint64_t v = labels[x + sx * y];
			if (x > 0 && ((v == labels[(x - 1) + sx * y]) == (permissible != 0))) {
				adj[x + sxe * y] |= 0x4;
				adj[x + sxe * (y + 1)] |= 0x8;
				any = 1;
			}
			if (y > 0 && ((v == labels[x + sx * (y - 1)]) == (permissible != 0))) {
				adj[x + sxe * y] |= 0x1;
				adj[(x + 1) + sxe * y] |= 0x2;
				any = 1;
			}
		}
	}
	cs->chains = NULL; cs->n = cs->cap = 0;
	if (!any) { free(adj); return; }

	const int64_t dir[4] = { 1, -1, sxe, -sxe };
	static const char sym[4] = { 'r', 'l', 'd', 'u' };
	static const uint8_t clear_fwd[4] = { 0xE, 0xD, 0xB, 0x7 };  /* bit k off */
	static const uint8_t clear_rev[4] = { 0xD, 0xE, 0x7, 0xB };  /* opposite bit off at the far end */

	int64_t* revisit = NULL; size_t rn = 0, rcap = 0;
	unsigned char* code = NULL; size_t cn = 0, ccap = 0;
	int64_t start = 0;
	const int64_t nverts = sxe * sye;

	for (;;) {
		while (start < nverts && !adj[start]) start++;   /* next_cluster (41-49) */
		if (start >= nverts) break;
		int64_t node = start;
		cn = 0; rn = 0;
		int64_t branches_taken = 1;
		while (adj[node] || rn) {
			if (cn + 4 > ccap) { ccap = ccap ? ccap * 2 : 1024; code = (unsigned char*)xrealloc(code, ccap); }
			if (!adj[node]) {
				code[cn++] = 't';
				branches_taken--;
				node = revisit[--rn];
				continue;
			}
			else if (__builtin_popcount(adj[node]) > 1) {
				code[cn++] = 'b';
				if (rn == rcap) { rcap = rcap ? rcap * 2 : 256; revisit = (int64_t*)xrealloc(revisit, sizeof(int64_t) * rcap); }
				revisit[rn++] = node;
				branches_taken++;
			}
			const int k = __builtin_ctz(adj[node]);
			const int64_t next = node + dir[k];
			code[cn++] = (unsigned char)sym[k];
			adj[node] &= clear_fwd[k];
			adj[next] &= clear_rev[k];
			node = next;
		}
		while (branches_taken > 0) {
			if (cn + 1 > ccap) { ccap = ccap ? ccap * 2 : 1024; code = (unsigned char*)xrealloc(code, ccap); }
			code[cn++] = 't';
			branches_taken--;
		}
		const int64_t adjusted = remove_initial_branch(start, code, cn, sx);
		remove_spurious_branches(code, cn);

		if (cs->n == cs->cap) { cs->cap = cs->cap ? cs->cap * 2 : 8; cs->chains = (chain_t*)xrealloc(cs->chains, sizeof(chain_t) * cs->cap); }
		chain_t* c = &cs->chains[cs->n++];
		c->node = (uint64_t)adjusted;
		c->codes = (uint8_t*)xmalloc(cn * 2 + 2);
		c->n = symbols_to_codepoints(code, cn, c->codes);
	}
	free(adj); free(revisit); free(code);
	/* chains are consumed in ascending start-vertex order (pack_codepoints 460-464) */
	qsort(cs->chains, cs->n, sizeof(chain_t), chain_cmp);
}

/* write_boc_index (src/crackcodes.hpp:318-372); nodes ascending */
static void write_boc_index(const chainset_t* cs, uint64_t sx, uint64_t sy, bytes_t* out) {
	const uint64_t sxe = sx + 1;
	const int xw = byte_width(sx + 1), yw = byte_width(sy + 1);
	/* count distinct y */
	uint64_t index_size = (uint64_t)yw, num_y = 0;
	for (size_t i = 0; i < cs->n; ) {
		uint64_t y = cs->chains[i].node / sxe;
		size_t j = i;
		while (j < cs->n && cs->chains[j].node / sxe == y) j++;
		index_size += (uint64_t)yw + (uint64_t)(j - i + 1) * (uint64_t)xw;
		num_y++;
		i = j;
	}
	bput(out, index_size, 4);
	bput(out, num_y, yw);
	uint64_t last_y = 0;
	for (size_t i = 0; i < cs->n; ) {
		uint64_t y = cs->chains[i].node / sxe;
		size_t j = i;
		while (j < cs->n && cs->chains[j].node / sxe == y) j++;
		bput(out, y - last_y, yw);
		last_y = y;
		bput(out, (uint64_t)(j - i), xw);
		uint64_t last_x = 0;
		for (size_t k = i; k < j; k++) {
			uint64_t x = cs->chains[k].node - sxe * y;
			bput(out, x - last_x, xw);
			last_x = x;
		}
		i = j;
	}
}
/* pack_codepoints (src/crackcodes.hpp:455-496) */
static void pack_codepoints(const chainset_t* cs, uint64_t sx, uint64_t sy, bytes_t* out) {
	write_boc_index(cs, sx, sy, out);
	uint8_t last = 0, encoded = 0;
	int pos = 0;
	for (size_t i = 0; i < cs->n; i++) {
		for (size_t k = 0; k < cs->chains[i].n; k++) {
			uint8_t cp = cs->chains[i].codes[k];
			uint8_t d = (uint8_t)((cp - last) & 3);
			last = cp;
			encoded |= (uint8_t)(d << pos);
			pos += 2;
			if (pos == 8) { bput8(out, encoded); encoded = 0; pos = 0; }
		}
	}
	if (pos > 0) bput8(out, encoded);
}

/* ------------------------------------------------------------------ */
/* markov coder (src/markov.hpp)                                       */
/* ------------------------------------------------------------------ */
static const uint8_t MK_LUT[24] = {
	/* permutations of (0,1,2,3) in lexicographic order, element i in bits 2i..2i+1
	 * (src/markov.hpp:20-68) */
	0xE4, 0xB4, 0xD8, 0x78, 0x9C, 0x6C, 0xE1, 0xB1, 0xC9, 0x39, 0x8D, 0x2D,
	0xD2, 0x72, 0xC6, 0x36, 0x4E, 0x1E, 0x93, 0x63, 0x87, 0x27, 0x4B, 0x1B
};
static int mk_ilut(uint8_t key) {
	for (int i = 0; i < 24; i++) if (MK_LUT[i] == key) return i;
	return 255;
}
/* difference_codepoints (src/markov.hpp:166-191): whole-slice mod-4 diff, first raw */
static uint8_t* slice_diffcodes(const chainset_t* cs, size_t* n_out) {
	size_t n = 0;
	for (size_t i = 0; i < cs->n; i++) n += cs->chains[i].n;
	uint8_t* d = (uint8_t*)xmalloc(n + 1);
	size_t m = 0;
	uint8_t last = 0;
	for (size_t i = 0; i < cs->n; i++) {
		for (size_t k = 0; k < cs->chains[i].n; k++) {
			uint8_t cp = cs->chains[i].codes[k];
			d[m] = (m == 0) ? cp : (uint8_t)((cp - last) & 3);
			last = cp;
			m++;
		}
	}
	*n_out = n;
	return d;
}
/* gather_statistics (src/markov.hpp:193-220) for one slice; ctx = CircularBuf value
 * (oldest code in the least-significant base-4 digit, src/markov.hpp:132-139) */
static void mk_stats_slice(const uint8_t* code, size_t n, int order, uint32_t* stats /* [4^order][4] */) {
	uint32_t ctx = 0;
	const int shift = 2 * (order - 1);
	for (size_t i = 0; i < n; i++) {
		__atomic_fetch_add(&stats[ctx * 4 + code[i]], 1u, __ATOMIC_RELAXED);
		ctx = (ctx >> 2) + ((uint32_t)code[i] << shift);
	}
}
/* stats_to_model (src/markov.hpp:222-266).  std::sort on 4 elements with the
 * non-strict comparator `a.count >= b.count` is libstdc++'s insertion sort:
 * each element is inserted in front of every earlier element whose count is <=
 * its own (count descending, ties -> larger symbol first). */
static void mk_stats_to_model(const uint32_t* stats, size_t rows, uint8_t* model /* [rows][4]: symbol -> rank */) {
	for (size_t r = 0; r < rows; r++) {
		int sym[4]; uint32_t cnt[4];
		int m = 0;
		for (int l = 0; l < 4; l++) {
			uint32_t c = stats[r * 4 + l];
			int p = m;
			while (p > 0 && c >= cnt[p - 1]) { sym[p] = sym[p - 1]; cnt[p] = cnt[p - 1]; p--; }
			sym[p] = l; cnt[p] = c;
			m++;
		}
		for (int j = 0; j < 4; j++) model[r * 4 + sym[j]] = (uint8_t)j;
	}
}
/* to_stored_model (src/markov.hpp:325-380) */
static int mk_to_stored(const uint8_t* model, size_t rows, bytes_t* out) {
	int pos = 0;
	uint32_t acc = 0;
	for (size_t r = 0; r < rows; r++) {
		uint8_t key = 0;
		for (int s = 0; s < 4; s++) key |= (uint8_t)(s << (2 * model[r * 4 + s]));   /* rank j holds symbol s */
		int idx = mk_ilut(key);
		if (idx == 255) FAIL("Corrupted model.");
		acc |= ((uint32_t)idx << pos);
		pos += 5;
		if (pos > 8) { bput8(out, (uint8_t)acc); pos -= 8; acc >>= 8; }
	}
	if (pos > 0) bput8(out, (uint8_t)acc);
	return 0;
}
/* from_stored_model (src/markov.hpp:382-420): rows as rank -> symbol */
static uint8_t* mk_from_stored(const unsigned char* stream, uint64_t nbytes, int order) {
	size_t rows = (size_t)1 << (2 * order);
	uint8_t* model = (uint8_t*)xcalloc(rows * 4, 1);
	for (size_t r = 0; r < rows; r++) {
		uint64_t bit = (uint64_t)r * 5;
		uint64_t byte = bit >> 3;
		int pos = (int)(bit & 7);
		uint32_t v = 0;
		if (byte < nbytes) v = stream[byte];
		if (byte + 1 < nbytes) v |= ((uint32_t)stream[byte + 1]) << 8;
		uint32_t decoded = (v >> pos) & 31;
		uint8_t row = decoded < 24 ? MK_LUT[decoded] : 0;
		model[r * 4 + 0] = row & 3;
		model[r * 4 + 1] = (row >> 2) & 3;
		model[r * 4 + 2] = (row >> 4) & 3;
		model[r * 4 + 3] = (row >> 6) & 3;
	}
	return model;
}
/* encode_markov (src/markov.hpp:422-473) */
static void mk_encode(const uint8_t* code, size_t n, const uint8_t* model, int order, bytes_t* out) {
	if (n == 0) return;
	int pos = 2;
	uint32_t acc = code[0];
	const int shift = 2 * (order - 1);
	uint32_t ctx = (uint32_t)code[0] << shift;
	for (size_t i = 1; i < n; i++) {
		uint8_t idx = model[ctx * 4 + code[i]];
		if (idx == 0) pos += 1;
		else if (idx == 1) { acc |= (1u << pos); pos += 2; }
		else if (idx == 2) { acc |= (3u << pos); pos += 3; }
		else { acc |= (7u << pos); pos += 3; }
		if (pos >= 8) { bput8(out, (uint8_t)acc); pos -= 8; acc >>= 8; }
		ctx = (ctx >> 2) + ((uint32_t)code[i] << shift);
	}
	if (pos > 0) bput8(out, (uint8_t)acc);
}
/* decode_codepoints (src/markov.hpp:268-323): returns undiffed moves */
static uint8_t* mk_decode(const unsigned char* stream, uint64_t nbytes, const uint8_t* model /* rank->symbol */, int order, size_t* n_out) {
	uint8_t* out = (uint8_t*)xmalloc((size_t)nbytes * 8 + 8);
	size_t m = 0;
	if (nbytes == 0) { *n_out = 0; return out; }   /* reference reads stream[0] out of bounds here */
	const int shift = 2 * (order - 1);
	uint8_t start = stream[0] & 3;
	out[m++] = start;
	uint32_t ctx = (uint32_t)start << shift;
	int pos = 2;
	for (uint64_t i = 0; i < nbytes; i++) {
		uint32_t byte = stream[i];
		if (i + 1 < nbytes) byte |= ((uint32_t)stream[i + 1]) << 8;
		while (pos < 8) {
			uint32_t cp = (byte >> pos) & 7;
			uint8_t v;
			if ((cp & 1) == 0) { v = model[ctx * 4 + 0]; pos += 1; }
			else if ((cp & 2) == 0) { v = model[ctx * 4 + 1]; pos += 2; }
			else if ((cp & 4) == 0) { v = model[ctx * 4 + 2]; pos += 3; }
			else { v = model[ctx * 4 + 3]; pos += 3; }
			out[m++] = v;
			ctx = (ctx >> 2) + ((uint32_t)v << shift);
		}
		pos -= 8;
	}
	for (size_t i = 1; i < m; i++) out[i] = (uint8_t)((out[i] + out[i - 1]) & 3);
	*n_out = m;
	return out;
}

/* ------------------------------------------------------------------ */
/* compress (src/crackle.hpp:34-257)                                   */
/* ------------------------------------------------------------------ */
typedef struct {
	const uint64_t* labels;
	int64_t sx, sy, sz;
	int permissible;
	chainset_t* chains;      /* [sz] */
	bytes_t* codes;          /* [sz] */
	int order;
	const uint8_t* model;    /* symbol -> rank */
	uint32_t* stats;
	/* flat labels */
	uint32_t** cc_scratch; uint32_t** ids_scratch; uint32_t** ren_scratch;
	uint64_t** mapping; uint64_t* ncomp; uint32_t* crcs;
} enc_ctx_t;

static void enc_boundaries_task(int64_t z, size_t tid, void* c) {
	enc_ctx_t* e = (enc_ctx_t*)c;
	create_crack_codes(e->labels + e->sx * e->sy * z, e->sx, e->sy, e->permissible, &e->chains[z]);
}
static void enc_pack_task(int64_t z, size_t tid, void* c) {
	enc_ctx_t* e = (enc_ctx_t*)c;
	pack_codepoints(&e->chains[z], (uint64_t)e->sx, (uint64_t)e->sy, &e->codes[z]);
}
static void enc_stats_task(int64_t z, size_t tid, void* c) {
	enc_ctx_t* e = (enc_ctx_t*)c;
	size_t n;
	uint8_t* d = slice_diffcodes(&e->chains[z], &n);
	mk_stats_slice(d, n, e->order, e->stats);
	free(d);
}
static void enc_markov_task(int64_t z, size_t tid, void* c) {
	/* markov::compress (src/markov.hpp:475-489) */
	enc_ctx_t* e = (enc_ctx_t*)c;
	size_t n;
	uint8_t* d = slice_diffcodes(&e->chains[z], &n);
	write_boc_index(&e->chains[z], (uint64_t)e->sx, (uint64_t)e->sy, &e->codes[z]);
	mk_encode(d, n, e->model, e->order, &e->codes[z]);
	free(d);
}
/* encode_flat per-slice part (src/labels.hpp:56-88) */
static void enc_flat_task(int64_t z, size_t tid, void* c) {
	enc_ctx_t* e = (enc_ctx_t*)c;
	const int64_t sxy = e->sx * e->sy;
	uint32_t* cc = e->cc_scratch[tid];
	const uint64_t* lab = e->labels + sxy * z;
	uint64_t N = ccl_labels_slice(lab, e->sx, e->sy, cc, 0, e->ids_scratch[tid], e->ren_scratch[tid]);
	uint64_t* mapping = (uint64_t*)xmalloc(sizeof(uint64_t) * (N + 1));
	uint32_t last = cc[0];
	mapping[cc[0]] = lab[0];
	for (int64_t i = 1; i < sxy; i++) {
		if (cc[i] != last) {
			mapping[cc[i]] = lab[i];
			last = cc[i];
		}
	}
	e->mapping[z] = mapping;
	e->ncomp[z] = N;
	e->crcs[z] = ckl_oracle_crc32c((const uint8_t*)cc, (uint64_t)sxy * 4);
}
static int u64_cmp(const void* a, const void* b) {
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

static int pins_encode(
	const uint64_t* labels, int64_t sx, int64_t sy, int64_t sz,
	const header_t* head, int stored_width, int auto_bgcolor, int64_t manual_bgcolor,
	bytes_t* labels_binary, uint32_t* crcs);

/* Sharded-encode hooks (test infrastructure for crackle_amd/distributed.py): the
 * whole-volume decisions of compress (src/crackle.hpp:48-64, 107-130, 233-235) can be
 * imposed from outside, exactly like ckl_encode_overrides in include/crackle_amd.h.
 * force_crack_format / force_label_format < 0 and force_stored_width == 0 mean "decide
 * from this volume"; model (4^order x 4 bytes, symbol -> rank) may be NULL. */
int ckl_oracle_compress_ex(
	const void* labels_v, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel,
	int force_crack_format, int force_label_format, int force_stored_width, const uint8_t* forced_model,
	unsigned char** out, uint64_t* out_len);

int ckl_oracle_compress(
	const void* labels_v, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel, unsigned char** out, uint64_t* out_len
) {
	return ckl_oracle_compress_ex(labels_v, dtype_bytes, is_signed, sx, sy, sz, allow_pins, fortran_order,
		markov_order, optimize_pins, auto_bgcolor, manual_bgcolor, parallel, -1, -1, 0, NULL, out, out_len);
}

int ckl_oracle_compress_ex(
	const void* labels_v, int dtype_bytes, int is_signed,
	int64_t sx, int64_t sy, int64_t sz,
	int allow_pins, int fortran_order, uint64_t markov_order,
	int optimize_pins, int auto_bgcolor, int64_t manual_bgcolor,
	uint64_t parallel,
	int force_crack_format, int force_label_format, int force_stored_width, const uint8_t* forced_model,
	unsigned char** out, uint64_t* out_len
) {
	if (is_signed) FAIL("ckl_oracle: signed labels are rejected by crackle.compress (crackle/codec.py:720-721)");
	if (optimize_pins) FAIL("ckl_oracle: allow_pins=2 (find_optimal_pins) is out of scope (SURVEY.md section 2 row 6)");
	if (dtype_bytes != 1 && dtype_bytes != 2 && dtype_bytes != 4 && dtype_bytes != 8) FAIL("ckl_oracle: bad dtype width");
	const int64_t voxels = sx * sy * sz;
	const int64_t sxy = sx * sy;
	uint64_t* labels = widen(labels_v, dtype_bytes, (uint64_t)voxels);

	/* max_label + compute_byte_width (src/lib.hpp:224-247) */
	uint64_t mx = 0;
	for (int64_t i = 0; i < voxels; i++) if (labels[i] > mx) mx = labels[i];
	const int stored_width = force_stored_width ? force_stored_width : byte_width(mx);
	/* pixel_pairs (src/lib.hpp:249-256) */
	uint64_t num_pairs = 0;
	for (int64_t i = 1; i < voxels; i++) num_pairs += (labels[i] == labels[i - 1]);

	header_t head;
	memset(&head, 0, sizeof head);
	head.format_version = 1;
	head.crack_format = IMPERMISSIBLE;
	head.label_format = PINS_VARIABLE_WIDTH;
	if ((int64_t)num_pairs < voxels / 2) {
		head.crack_format = PERMISSIBLE;
		head.label_format = FLAT;
	}
	if (force_crack_format >= 0) {
		head.crack_format = force_crack_format;
		head.label_format = force_crack_format == PERMISSIBLE ? FLAT : PINS_VARIABLE_WIDTH;
	}
	if (sz == 1 || !allow_pins) head.label_format = FLAT;
	if (force_label_format >= 0) head.label_format = force_label_format;
	head.is_signed = 0;
	head.data_width = dtype_bytes;
	head.stored_data_width = stored_width;
	head.sx = (uint32_t)sx; head.sy = (uint32_t)sy; head.sz = (uint32_t)sz;
	head.log2_grid_size = 31;
	head.num_label_bytes = 0;
	head.fortran_order = fortran_order;
	head.markov_model_order = (int)(markov_order & 0xFF);
	head.is_sorted = 1;

	bytes_t final = { 0 };
	if (voxels == 0) {
		header_write(&head, &final);
		*out = final.p; *out_len = final.n;
		free(labels);
		return 0;
	}

	const size_t threads = resolve_parallel(parallel, sz);
	enc_ctx_t e;
	memset(&e, 0, sizeof e);
	e.labels = labels; e.sx = sx; e.sy = sy; e.sz = sz;
	e.permissible = (head.crack_format == PERMISSIBLE);
	e.chains = (chainset_t*)xcalloc((size_t)sz, sizeof(chainset_t));
	e.codes = (bytes_t*)xcalloc((size_t)sz, sizeof(bytes_t));

	parallel_for(sz, threads, enc_boundaries_task, &e);   /* encode_boundaries (crackcodes.hpp:498-521) */

	if (head.markov_model_order > 0) {   /* src/crackle.hpp:107-118 */
		int empty = 1;
		for (int64_t z = 0; z < sz; z++) if (e.chains[z].n) { empty = 0; break; }
		if (empty && !forced_model) head.markov_model_order = 0;
	}

	bytes_t stored_model = { 0 };
	int rc = 0;
	if (head.markov_model_order > 0) {
		const int order = head.markov_model_order;
		const size_t rows = (size_t)1 << (2 * order);
		e.order = order;
		e.stats = (uint32_t*)xcalloc(rows * 4, sizeof(uint32_t));
		uint8_t* model = (uint8_t*)xmalloc(rows * 4);
		if (forced_model) memcpy(model, forced_model, rows * 4);
		else {
			parallel_for(sz, threads, enc_stats_task, &e);
			mk_stats_to_model(e.stats, rows, model);
		}
		rc = mk_to_stored(model, rows, &stored_model);
		e.model = model;
		if (!rc) parallel_for(sz, threads, enc_markov_task, &e);
		free(model);
		free(e.stats);
	}
	else {
		parallel_for(sz, threads, enc_pack_task, &e);
	}

	bytes_t labels_binary = { 0 };
	uint32_t* crcs = (uint32_t*)xcalloc((size_t)sz, sizeof(uint32_t));
	if (!rc && head.label_format == PINS_VARIABLE_WIDTH) {
		rc = pins_encode(labels, sx, sy, sz, &head, stored_width, auto_bgcolor, manual_bgcolor, &labels_binary, crcs);
	}
	else if (!rc) {
		/* encode_flat (src/labels.hpp:30-155) */
		e.cc_scratch = (uint32_t**)xcalloc(threads, sizeof(uint32_t*));
		e.ids_scratch = (uint32_t**)xcalloc(threads, sizeof(uint32_t*));
		e.ren_scratch = (uint32_t**)xcalloc(threads, sizeof(uint32_t*));
		for (size_t t = 0; t < threads; t++) {
			e.cc_scratch[t] = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)sxy);
			e.ids_scratch[t] = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(sxy + 2));
			e.ren_scratch[t] = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(sxy + 2));
		}
		e.mapping = (uint64_t**)xcalloc((size_t)sz, sizeof(uint64_t*));
		e.ncomp = (uint64_t*)xcalloc((size_t)sz, sizeof(uint64_t));
		e.crcs = crcs;
		parallel_for(sz, threads, enc_flat_task, &e);
		uint64_t N = 0;
		for (int64_t z = 0; z < sz; z++) N += e.ncomp[z];
		uint64_t* mapping = (uint64_t*)xmalloc(sizeof(uint64_t) * (N + 1));
		uint64_t* uniq = (uint64_t*)xmalloc(sizeof(uint64_t) * (N + 1));
		uint64_t k = 0;
		for (int64_t z = 0; z < sz; z++) {
			memcpy(mapping + k, e.mapping[z], sizeof(uint64_t) * e.ncomp[z]);
			k += e.ncomp[z];
			free(e.mapping[z]);
		}
		memcpy(uniq, mapping, sizeof(uint64_t) * N);
		qsort(uniq, N, sizeof(uint64_t), u64_cmp);
		uint64_t nu = N ? 1 : 0;
		for (uint64_t i = 1; i < N; i++) if (uniq[i] != uniq[i - 1]) uniq[nu++] = uniq[i];
		const int key_width = byte_width(nu);
		const int component_width = byte_width((uint64_t)sxy);
		bput(&labels_binary, nu, 8);
		for (uint64_t i = 0; i < nu; i++) bput(&labels_binary, uniq[i], stored_width);
		for (int64_t z = 0; z < sz; z++) bput(&labels_binary, e.ncomp[z], component_width);
		for (uint64_t i = 0; i < N; i++) {
			/* remapping[mapping[i]] = index in uniq */
			uint64_t lo = 0, hi = nu;
			while (lo + 1 < hi) { uint64_t mid = (lo + hi) / 2; if (uniq[mid] <= mapping[i]) lo = mid; else hi = mid; }
			bput(&labels_binary, lo, key_width);
		}
		free(mapping); free(uniq);
		for (size_t t = 0; t < threads; t++) { free(e.cc_scratch[t]); free(e.ids_scratch[t]); free(e.ren_scratch[t]); }
		free(e.cc_scratch); free(e.ids_scratch); free(e.ren_scratch); free(e.mapping); free(e.ncomp);
	}

	if (!rc) {
		/* assembly (src/crackle.hpp:171-216) */
		head.num_label_bytes = labels_binary.n;
		header_write(&head, &final);
		size_t zi = final.n;
		for (int64_t z = 0; z < sz; z++) bput(&final, (uint32_t)e.codes[z].n, 4);
		bput(&final, ckl_oracle_crc32c(final.p + zi, (uint64_t)sz * 4), 4);
		bpush(&final, labels_binary.p, labels_binary.n);
		if (head.markov_model_order > 0) bpush(&final, stored_model.p, stored_model.n);
		for (int64_t z = 0; z < sz; z++) bpush(&final, e.codes[z].p, e.codes[z].n);
		bput(&final, ckl_oracle_crc32c(labels_binary.p, labels_binary.n), 4);
		for (int64_t z = 0; z < sz; z++) bput(&final, crcs[z], 4);
	}

	for (int64_t z = 0; z < sz; z++) { chainset_free(&e.chains[z]); free(e.codes[z].p); }
	free(e.chains); free(e.codes); free(crcs); free(labels_binary.p); free(stored_model.p); free(labels);
	if (rc) { free(final.p); return rc; }
	*out = final.p; *out_len = final.n;
	return 0;
}

/* ------------------------------------------------------------------ */
/* decode: crack code -> symbols -> VCG                                 */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t node; size_t begin, end; } dchain_t;   /* symbols[begin,end) */

/* read_boc_index (src/crackcodes.hpp:283-316) */
static uint64_t* read_boc_index(const unsigned char* code, uint64_t code_len, uint64_t sx, uint64_t sy, size_t* n_out, int* bad) {
	const uint64_t sxe = sx + 1;
	const int xw = byte_width(sx + 1), yw = byte_width(sy + 1);
	*n_out = 0; *bad = 0;
	if (code_len < 4 + (uint64_t)yw) { *bad = 1; return NULL; }
	uint64_t idx = 4;
	uint64_t num_y = rd(code, idx, yw); idx += yw;
	size_t cap = 16, n = 0;
	uint64_t* nodes = (uint64_t*)xmalloc(sizeof(uint64_t) * cap);
	uint64_t y = 0;
	for (uint64_t yi = 0; yi < num_y; yi++) {
		if (idx + yw + xw > code_len) { *bad = 1; break; }
		y += rd(code, idx, yw); idx += yw;
		uint64_t num_x = rd(code, idx, xw); idx += xw;
		uint64_t x = 0;
		for (uint64_t xi = 0; xi < num_x; xi++) {
			if (idx + xw > code_len) { *bad = 1; break; }
			x += rd(code, idx, xw); idx += xw;
			if (n == cap) { cap *= 2; nodes = (uint64_t*)xrealloc(nodes, sizeof(uint64_t) * cap); }
			nodes[n++] = x + sxe * y;
		}
		if (*bad) break;
	}
	*n_out = n;
	return nodes;
}

/* packed_codepoints_to_symbols / codepoints_to_symbols
 * (src/crackcodes.hpp:523-676): 5-state FSM */
static dchain_t* moves_to_symbols(const uint8_t* moves, size_t n_moves, const uint64_t* nodes, size_t n_nodes, unsigned char* symbols, size_t* n_chains_out) {
	static const char remap[4] = { 'u', 'r', 'd', 'l' };
	dchain_t* chains = (dchain_t*)xmalloc(sizeof(dchain_t) * (n_nodes + 1));
	size_t nc = 0, ns = 0, chain_begin = 0;
	uint64_t branches_taken = 0;
	size_t node_i = 0;
	uint64_t node = 0;
	uint8_t last_move = 255;
	for (size_t i = 0; i < n_moves; i++) {
		if (branches_taken == 0) {
			if (node_i >= n_nodes) break;
			node = nodes[node_i++];
			branches_taken = 1;
			chain_begin = ns;
		}
		uint8_t move = moves[i];
		if ((move ^ last_move) != 2) {
			symbols[ns++] = (unsigned char)remap[move];
			last_move = move;
			continue;
		}
		else if (move == 0 || move == 3) {   /* popcount(move) != 1 */
			symbols[ns - 1] = 't';
			branches_taken--;
			last_move = 255;
		}
		else {
			symbols[ns - 1] = 'b';
			branches_taken++;
			last_move = 255;
		}
		if (branches_taken == 0) {
			chains[nc].node = node; chains[nc].begin = chain_begin; chains[nc].end = ns;
			nc++;
		}
	}
	*n_chains_out = nc;
	return chains;
}

/* decode_(im)permissible_crack_code (src/crackcodes.hpp:706-876) */
static int rasterize(const dchain_t* chains, size_t nc, const unsigned char* symbols, int64_t sx, int64_t sy, int permissible, uint8_t* edges) {
	memset(edges, permissible ? 0 : 0xF, (size_t)(sx * sy));
	const int64_t sxe = sx + 1;
	const int64_t npix = sx * sy;
	int64_t* stack = NULL; size_t sp = 0, scap = 0;
#define TOUCH(idx, bit) do { int64_t _i = (idx); if (_i >= 0 && _i < npix) { if (permissible) edges[_i] |= (uint8_t)(bit); else edges[_i] &= (uint8_t)~(bit); } } while (0)
	for (size_t c = 0; c < nc; c++) {
		int64_t y = (int64_t)(chains[c].node / (uint64_t)sxe);
		int64_t x = (int64_t)(chains[c].node - (uint64_t)(sxe * y));
		int64_t loc = x + sx * y;
		sp = 0;
		for (size_t i = chains[c].begin; i < chains[c].end; i++) {
			unsigned char s = symbols[i];
			if (loc < 0 || loc >= (sx + 1) * (sy + 1)) { free(stack); return 1; }
			if (s == 'u') {
				if (x > 0 && y > 0) TOUCH(loc - 1 - sx, 0x1);
				if (y > 0) TOUCH(loc - sx, 0x2);
				y--; loc -= sx;
			}
			else if (s == 'd') {
				if (x > 0) TOUCH(loc - 1, 0x1);
				TOUCH(loc, 0x2);
				y++; loc += sx;
			}
			else if (s == 'l') {
				if (x > 0 && y > 0) TOUCH(loc - 1 - sx, 0x4);
				if (x > 0) TOUCH(loc - 1, 0x8);
				x--; loc--;
			}
			else if (s == 'r') {
				if (y > 0) TOUCH(loc - sx, 0x4);
				TOUCH(loc, 0x8);
				x++; loc++;
			}
			else if (s == 'b') {
				if (sp == scap) { scap = scap ? scap * 2 : 256; stack = (int64_t*)xrealloc(stack, sizeof(int64_t) * scap); }
				stack[sp++] = loc;
			}
			else if (s == 't') {
				if (sp) {
					loc = stack[--sp];
					y = loc / sx;
					x = loc - sx * y;
				}
			}
		}
	}
#undef TOUCH
	free(stack);
	return 0;
}

/* crack_code_to_vcg (src/crackle.hpp:393-425) */
static int slice_to_vcg(const unsigned char* code, uint64_t code_len, int64_t sx, int64_t sy, int permissible, const uint8_t* model, int order, uint8_t* vcg) {
	size_t n_nodes; int bad;
	uint64_t* nodes = read_boc_index(code, code_len, (uint64_t)sx, (uint64_t)sy, &n_nodes, &bad);
	if (bad) { free(nodes); return 1; }
	uint64_t index_size = 4 + rd(code, 0, 4);
	if (index_size > code_len) { free(nodes); return 1; }
	uint8_t* moves; size_t n_moves;
	if (order == 0) {
		/* unpack + undiff (src/crackcodes.hpp:547-563) */
		n_moves = (size_t)(code_len - index_size) * 4;
		moves = (uint8_t*)xmalloc(n_moves + 1);
		uint8_t last = 0;
		size_t m = 0;
		for (uint64_t i = index_size; i < code_len; i++) {
			for (int j = 0; j < 4; j++) {
				uint8_t mv = (uint8_t)(((code[i] >> (2 * j)) & 3) + last) & 3;
				last = mv;
				moves[m++] = mv;
			}
		}
	}
	else {
		moves = mk_decode(code + index_size, code_len - index_size, model, order, &n_moves);
	}
	unsigned char* symbols = (unsigned char*)xmalloc(n_moves + 1);
	size_t nc;
	dchain_t* chains = moves_to_symbols(moves, n_moves, nodes, n_nodes, symbols, &nc);
	int rc = rasterize(chains, nc, symbols, sx, sy, permissible, vcg);
	free(chains); free(symbols); free(moves); free(nodes);
	return rc;
}

/* ------------------------------------------------------------------ */
/* decompress (src/crackle.hpp:262-336, 447-663; labels.hpp:424-648)   */
/* ------------------------------------------------------------------ */
typedef struct {
	header_t head;
	const unsigned char* buf; uint64_t n;
	const unsigned char* labels_binary;
	uint64_t* z_index;            /* [sz+1] absolute offsets */
	uint8_t* model; int order;
	int64_t z_start, z_end;
	void* out; int has_label; uint64_t label;
	uint8_t** vcg; uint32_t** cc; uint32_t** ids; uint32_t** ren;
	/* label section, parsed once */
	uint64_t* comp_per_slice; uint64_t* comp_offset;   /* [sz], [sz+1] */
	/* flat */
	uint64_t keys_offset; int key_width; uint64_t num_unique; uint64_t uniq_offset;
	/* pins */
	uint64_t bgcolor;
	uint64_t n_pins; uint64_t* pin_label; uint64_t* pin_index; uint64_t* pin_depth;
	uint64_t n_ccl; uint64_t* ccl_label; uint64_t* ccl_id;
	/* captures for array_equal: per-voxel component ids, component counts and component -> label tables */
	uint32_t* cap_cc; uint64_t* cap_N; uint64_t** cap_lmap;
	atomic_int failed;
	char err[256];
} dec_ctx_t;

static uint64_t read_stored(const dec_ctx_t* d, uint64_t offset) {
	/* static_cast<LABEL>(STORED_LABEL): sign-extend when the stream is signed */
	const int w = d->head.stored_data_width;
	uint64_t v = rd(d->labels_binary, offset, w);
	if (d->head.is_signed && w < 8 && (v >> (8 * w - 1))) v |= ~0ull << (8 * w);
	return v;
}
static uint64_t load_label(const void* in, int width, uint64_t idx) {
	if (width == 1) return ((const uint8_t*)in)[idx];
	if (width == 2) return ((const uint16_t*)in)[idx];
	if (width == 4) return ((const uint32_t*)in)[idx];
	return ((const uint64_t*)in)[idx];
}
static void store_out(void* out, int width, uint64_t idx, uint64_t v) {
	if (width == 1) ((uint8_t*)out)[idx] = (uint8_t)v;
	else if (width == 2) ((uint16_t*)out)[idx] = (uint16_t)v;
	else if (width == 4) ((uint32_t*)out)[idx] = (uint32_t)v;
	else ((uint64_t*)out)[idx] = v;
}

static void dec_slice_task(int64_t zi, size_t tid, void* c) {
	dec_ctx_t* d = (dec_ctx_t*)c;
	const header_t* h = &d->head;
	const int64_t sx = h->sx, sy = h->sy, sxy = sx * sy;
	const int64_t z = d->z_start + zi;
	const int64_t szr = d->z_end - d->z_start;
	uint8_t* vcg = d->vcg[tid];
	uint32_t* cc = d->cc[tid];
	const unsigned char* code = d->buf + d->z_index[z];
	const uint64_t code_len = d->z_index[z + 1] - d->z_index[z];

	if (slice_to_vcg(code, code_len, sx, sy, h->crack_format == PERMISSIBLE, d->model, d->order, vcg)) {
		if (!atomic_exchange(&d->failed, 1)) snprintf(d->err, sizeof d->err, "crackle: malformed crack code on z=%lld", (long long)z);
		return;
	}
	const uint64_t N = ccl_vcg_slice(vcg, sx, sy, cc, d->ids[tid], d->ren[tid]);

	if (h->format_version > 0) {
		const uint32_t computed = ckl_oracle_crc32c((const uint8_t*)cc, (uint64_t)sxy * 4);
		const uint32_t stored = (uint32_t)rd(d->buf, d->n - (uint64_t)h->sz * 4 + (uint64_t)z * 4, 4);
		if (computed != stored) {
			/* the reference throws inside a pool task and the exception is swallowed
			 * (SURVEY Q8); the restatement reports it */
			if (!atomic_exchange(&d->failed, 1)) snprintf(d->err, sizeof d->err, "crackle: crack code crc mismatch on z=%lld computed: %u stored: %u", (long long)z, computed, stored);
			return;
		}
	}

	/* decode_label_map (src/labels.hpp:453-648) */
	uint64_t* label_map = (uint64_t*)xmalloc(sizeof(uint64_t) * (N + 1));
	if (h->label_format == FLAT) {
		if (N != d->comp_per_slice[z]) {
			if (!atomic_exchange(&d->failed, 1)) snprintf(d->err, sizeof d->err, "crackle: component count mismatch on z=%lld", (long long)z);
			free(label_map);
			return;
		}
		for (uint64_t i = 0; i < N; i++) {
			uint64_t key = rd(d->labels_binary, d->keys_offset + (d->comp_offset[z] + i) * (uint64_t)d->key_width, d->key_width);
			label_map[i] = key < d->num_unique ? read_stored(d, d->uniq_offset + key * (uint64_t)h->stored_data_width) : 0;
		}
	}
	else {
		for (uint64_t i = 0; i < N; i++) label_map[i] = d->bgcolor;
		const uint64_t left = d->comp_offset[z], right = d->comp_offset[z + 1];
		for (uint64_t j = 0; j < d->n_ccl; j++) {
			uint64_t id = d->ccl_id[j];
			if (id < left || id >= right) continue;
			if (id - left < N) label_map[id - left] = d->ccl_label[j];
		}
		for (uint64_t j = 0; j < d->n_pins; j++) {
			int64_t pin_z = (int64_t)(d->pin_index[j] / (uint64_t)sxy);
			int64_t loc = (int64_t)(d->pin_index[j] - (uint64_t)pin_z * (uint64_t)sxy);
			int64_t pin_z_end = pin_z + (int64_t)d->pin_depth[j] + 1;
			if (z >= pin_z && z < pin_z_end) {
				uint32_t id = cc[loc];
				if (id < N) label_map[id] = d->pin_label[j];
			}
		}
	}

	if (d->cap_cc) {
		memcpy(d->cap_cc + (uint64_t)zi * (uint64_t)sxy, cc, (size_t)sxy * 4);
		d->cap_N[zi] = N;
		d->cap_lmap[zi] = label_map;      /* ownership moves to the caller */
		return;
	}

	/* paint (src/crackle.hpp:617-656) */
	const int ow = d->has_label ? 1 : h->data_width;
	if (h->fortran_order) {
		for (int64_t i = 0; i < sxy; i++) {
			uint64_t v = label_map[cc[i]];
			if (d->has_label) v = (v == d->label);
			store_out(d->out, ow, (uint64_t)(i + zi * sxy), v);
		}
	}
	else {
		int64_t i = 0;
		for (int64_t y = 0; y < sy; y++) {
			for (int64_t x = 0; x < sx; x++, i++) {
				uint64_t v = label_map[cc[i]];
				if (d->has_label) v = (v == d->label);
				store_out(d->out, ow, (uint64_t)(zi + szr * (y + sy * x)), v);
			}
		}
	}
	free(label_map);
}

/* get_crack_code_offsets (src/crackle.hpp:262-313) */
static int dec_z_index(dec_ctx_t* d) {
	const header_t* h = &d->head;
	const uint64_t offset = header_bytes(h);
	if (offset + grid_index_bytes(h) > d->n) FAIL("crackle: get_crack_code_offsets: Unable to read past end of buffer.");
	if (h->format_version > 0) {
		uint32_t stored = (uint32_t)rd(d->buf, offset + 4ull * h->sz, 4);
		uint32_t computed = ckl_oracle_crc32c(d->buf + offset, 4ull * h->sz);
		if (stored != computed) FAIL("crackle: grid index crc32c did not match. stored: %u computed: %u", stored, computed);
	}
	d->z_index = (uint64_t*)xcalloc((size_t)h->sz + 1, sizeof(uint64_t));
	uint64_t base = offset + grid_index_bytes(h) + h->num_label_bytes + markov_model_bytes(h);
	d->z_index[0] = base;
	for (uint64_t z = 0; z < h->sz; z++) d->z_index[z + 1] = d->z_index[z] + rd(d->buf, offset + 4 * z, 4);
	if (d->z_index[h->sz] > d->n) FAIL("crackle: get_crack_codes: Unable to read past end of buffer.");
	return 0;
}

/* decode_components + flat / condensed-pins section layout
 * (src/labels.hpp:424-451, 453-506, 508-617), parsed once for all slices */
static int dec_labels_section(dec_ctx_t* d) {
	const header_t* h = &d->head;
	const unsigned char* lb = d->labels_binary;
	const uint64_t nlb = h->num_label_bytes;
	const int sw = h->stored_data_width;
	const int component_width = byte_width((uint64_t)h->sx * h->sy);
	d->comp_per_slice = (uint64_t*)xcalloc((size_t)h->sz + 1, sizeof(uint64_t));
	d->comp_offset = (uint64_t*)xcalloc((size_t)h->sz + 2, sizeof(uint64_t));
	uint64_t offset;
	if (h->label_format == FLAT) {
		if (nlb < 8) FAIL("crackle: label section is malformed or corrupted.");
		d->num_unique = rd(lb, 0, 8);
		d->uniq_offset = 8;
		offset = 8 + (uint64_t)sw * d->num_unique;
	}
	else if (h->label_format == PINS_VARIABLE_WIDTH) {
		if (nlb < (uint64_t)sw + 8) FAIL("crackle: pin section is malformed or corrupted.");
		d->bgcolor = read_stored(d, 0);
		d->num_unique = rd(lb, (uint64_t)sw, 8);
		d->uniq_offset = (uint64_t)sw + 8;
		offset = 8 + (uint64_t)sw * (d->num_unique + 1);
	}
	else FAIL("crackle: Unsupported label format. Got: %d", h->label_format);
	if (offset + (uint64_t)component_width * h->sz > nlb) FAIL("crackle: label section is malformed or corrupted.");
	for (uint64_t z = 0; z < h->sz; z++) {
		d->comp_per_slice[z] = rd(lb, offset + z * (uint64_t)component_width, component_width);
		d->comp_offset[z + 1] = d->comp_offset[z] + d->comp_per_slice[z];
	}
	offset += (uint64_t)component_width * h->sz;
	if (h->label_format == FLAT) {
		d->key_width = byte_width(d->num_unique);
		d->keys_offset = offset;
		if (offset + d->comp_offset[h->sz] * (uint64_t)d->key_width > nlb) FAIL("crackle: label section is malformed or corrupted.");
		return 0;
	}
	/* condensed pins */
	if (offset + 1 > nlb) FAIL("crackle: pin section is malformed or corrupted.");
	const uint8_t combined = lb[offset++];
	const int npw = 1 << (combined & 3), dw = 1 << ((combined >> 2) & 3), ccw = 1 << ((combined >> 4) & 3);
	const int iw = pin_index_width(h);
	size_t pcap = 64, ccap = 64;
	d->pin_label = (uint64_t*)xmalloc(8 * pcap); d->pin_index = (uint64_t*)xmalloc(8 * pcap); d->pin_depth = (uint64_t*)xmalloc(8 * pcap);
	d->ccl_label = (uint64_t*)xmalloc(8 * ccap); d->ccl_id = (uint64_t*)xmalloc(8 * ccap);
	uint64_t i = offset;
	for (uint64_t label = 0; label < d->num_unique; label++) {
		if (i + (uint64_t)npw > nlb) FAIL("crackle: pin section is malformed or corrupted.");
		const uint64_t lv = read_stored(d, d->uniq_offset + label * (uint64_t)sw);
		uint64_t num_pins = rd(lb, i, npw); i += npw;
		if (i + num_pins * (uint64_t)(iw + dw) + (uint64_t)npw > nlb) FAIL("crackle: pin section is malformed or corrupted.");
		uint64_t idx = 0;
		for (uint64_t j = 0; j < num_pins; j++) {
			idx += rd(lb, i + j * (uint64_t)iw, iw);
			uint64_t depth = rd(lb, i + num_pins * (uint64_t)iw + j * (uint64_t)dw, dw);
			if (d->n_pins == pcap) {
				pcap *= 2;
				d->pin_label = (uint64_t*)xrealloc(d->pin_label, 8 * pcap);
				d->pin_index = (uint64_t*)xrealloc(d->pin_index, 8 * pcap);
				d->pin_depth = (uint64_t*)xrealloc(d->pin_depth, 8 * pcap);
			}
			d->pin_label[d->n_pins] = lv; d->pin_index[d->n_pins] = idx; d->pin_depth[d->n_pins] = depth;
			d->n_pins++;
		}
		i += num_pins * (uint64_t)(iw + dw);
		uint64_t num_cc = rd(lb, i, npw); i += npw;
		if (i + num_cc * (uint64_t)ccw > nlb) FAIL("crackle: pin section is malformed or corrupted.");
		uint64_t id = 0;
		for (uint64_t j = 0; j < num_cc; j++) {
			id += rd(lb, i, ccw); i += ccw;
			id &= 0xFFFFFFFFull;   /* std::vector<uint32_t> cc_labels (labels.hpp:580-587) */
			if (d->n_ccl == ccap) {
				ccap *= 2;
				d->ccl_label = (uint64_t*)xrealloc(d->ccl_label, 8 * ccap);
				d->ccl_id = (uint64_t*)xrealloc(d->ccl_id, 8 * ccap);
			}
			d->ccl_label[d->n_ccl] = lv; d->ccl_id[d->n_ccl] = id;
			d->n_ccl++;
		}
	}
	return 0;
}

static void dec_free(dec_ctx_t* d, size_t threads) {
	free(d->z_index); free(d->model); free(d->comp_per_slice); free(d->comp_offset);
	free(d->pin_label); free(d->pin_index); free(d->pin_depth); free(d->ccl_label); free(d->ccl_id);
	for (size_t t = 0; t < threads; t++) {
		if (d->vcg) free(d->vcg[t]);
		if (d->cc) free(d->cc[t]);
		if (d->ids) free(d->ids[t]);
		if (d->ren) free(d->ren[t]);
	}
	free(d->vcg); free(d->cc); free(d->ids); free(d->ren);
}

static int dec_run(
	dec_ctx_t* dp, const unsigned char* buf, uint64_t n, void* out,
	int64_t z_start, int64_t z_end, uint64_t parallel,
	int has_label, uint64_t label
);

int ckl_oracle_decompress(
	const unsigned char* buf, uint64_t n, void* out,
	int64_t z_start, int64_t z_end, uint64_t parallel,
	int has_label, uint64_t label
) {
	dec_ctx_t d;
	memset(&d, 0, sizeof d);
	return dec_run(&d, buf, n, out, z_start, z_end, parallel, has_label, label);
}

/* the decoder proper; with dp->cap_* set the slices are captured instead of painted */
static int dec_run(
	dec_ctx_t* dp, const unsigned char* buf, uint64_t n, void* out,
	int64_t z_start, int64_t z_end, uint64_t parallel,
	int has_label, uint64_t label
) {
	dec_ctx_t d = *dp;
	if (n < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream. Bytes: %llu", (unsigned long long)n);
	if (header_read(&d.head, buf, n)) return 1;
	const header_t* h = &d.head;
	/* range clamp (src/crackle.hpp:527-537) */
	int64_t zs = z_start, ze = z_end;
	if (zs > (int64_t)h->sz - 1) zs = (int64_t)h->sz - 1;
	if (zs < 0) zs = 0;
	ze = ze < 0 ? (int64_t)h->sz : ze;
	if (ze > (int64_t)h->sz) ze = h->sz;
	if (ze < 0) ze = 0;
	if (zs >= ze) FAIL("crackle: Invalid range: %lld - %lld", (long long)zs, (long long)ze);
	const uint64_t sxy = (uint64_t)h->sx * h->sy;
	if (sxy * (uint64_t)(ze - zs) == 0) return 0;

	d.buf = buf; d.n = n;
	d.z_start = zs; d.z_end = ze;
	d.out = out; d.has_label = has_label; d.label = label;
	size_t threads = 0;
	int rc = 0;
	if (header_bytes(h) + grid_index_bytes(h) + h->num_label_bytes + markov_model_bytes(h) + 4 * ((uint64_t)h->sz + 1) > n) {
		snprintf(g_err, sizeof g_err, "crackle: Unable to read past end of buffer.");
		return 1;
	}
	d.labels_binary = buf + header_bytes(h) + grid_index_bytes(h);
	d.order = h->markov_model_order;
	if (d.order > 0) {
		d.model = mk_from_stored(d.labels_binary + h->num_label_bytes, markov_model_bytes(h), d.order);
	}
	rc = dec_z_index(&d);
	if (!rc) rc = dec_labels_section(&d);
	if (!rc) {
		threads = resolve_parallel(parallel, ze - zs);
		d.vcg = (uint8_t**)xcalloc(threads, sizeof(void*));
		d.cc = (uint32_t**)xcalloc(threads, sizeof(void*));
		d.ids = (uint32_t**)xcalloc(threads, sizeof(void*));
		d.ren = (uint32_t**)xcalloc(threads, sizeof(void*));
		for (size_t t = 0; t < threads; t++) {
			d.vcg[t] = (uint8_t*)xmalloc(sxy);
			d.cc[t] = (uint32_t*)xmalloc(sxy * 4);
			d.ids[t] = (uint32_t*)xmalloc((sxy + 2) * 4);
			d.ren[t] = (uint32_t*)xmalloc((sxy + 2) * 4);
		}
		parallel_for(ze - zs, threads, dec_slice_task, &d);
		if (atomic_load(&d.failed)) {
			snprintf(g_err, sizeof g_err, "%s", d.err);
			rc = 1;
		}
	}
	dec_free(&d, threads);
	return rc;
}

/* operations::get_szr (src/operations.hpp:54-72): slices of the clamped range; an empty range is an error */
static int get_szr(const header_t* h, int64_t z_start, int64_t z_end, int64_t* zs_out, int64_t* szr) {
	int64_t zs = z_start, ze = z_end;
	const int64_t last = (int64_t)(uint32_t)(h->sz - 1u);      /* header.sz - 1 in 32-bit unsigned arithmetic */
	if (zs > last) zs = last;
	if (zs < 0) zs = 0;
	ze = ze < 0 ? (int64_t)h->sz : ze;
	if (ze > (int64_t)h->sz) ze = h->sz;
	if (ze < 0) ze = 0;
	if (zs >= ze) FAIL("crackle: Invalid range: %lld - %lld", (long long)zs, (long long)ze);
	*zs_out = zs; *szr = ze - zs;
	return 0;
}

/* operations::array_equal (src/operations.hpp:1039-1184), quirk included: label_map1 on both sides */
int ckl_oracle_array_equal(const unsigned char* buf1, uint64_t n1, const unsigned char* buf2, uint64_t n2, uint64_t parallel, int* equal) {
	header_t h1, h2;
	*equal = 0;
	if (n1 < HEADER_BYTES_V0 || n2 < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream.");
	if (header_read(&h1, buf1, n1) || header_read(&h2, buf2, n2)) return 1;
	int64_t zs1, zs2, szr1, szr2;
	if (get_szr(&h1, 0, -1, &zs1, &szr1) || get_szr(&h2, 0, -1, &zs2, &szr2)) return 1;      /* get_voxels throws for sz = 0 */
	const uint64_t v1 = (uint64_t)h1.sx * h1.sy * (uint64_t)szr1, v2 = (uint64_t)h2.sx * h2.sy * (uint64_t)szr2;
	if (v1 == 0 || v2 == 0) { *equal = v1 == v2; return 0; }                       /* :1049-1054 */
	if (h1.sx != h2.sx || h1.sy != h2.sy || h1.sz != h2.sz) return 0;              /* :1056-1062 */
	const uint64_t sxy = (uint64_t)h1.sx * h1.sy, sz = h1.sz;
	dec_ctx_t d[2];
	int rc = 0;
	for (int k = 0; k < 2 && !rc; k++) {
		memset(&d[k], 0, sizeof d[k]);
		d[k].cap_cc = (uint32_t*)xmalloc(sxy * sz * 4);
		d[k].cap_N = (uint64_t*)xcalloc(sz, sizeof(uint64_t));
		d[k].cap_lmap = (uint64_t**)xcalloc(sz, sizeof(uint64_t*));
		rc = dec_run(&d[k], k ? buf2 : buf1, k ? n2 : n1, NULL, 0, -1, parallel, 0, 0);
	}
	if (!rc) {
		int eq = 1;
		for (uint64_t z = 0; z < sz && eq; z++) {
			if (d[0].cap_N[z] != d[1].cap_N[z]) { eq = 0; break; }                     /* :1146-1149 */
			const uint64_t* lm1 = d[0].cap_lmap[z];
			const uint32_t* c1 = d[0].cap_cc + z * sxy;
			const uint32_t* c2 = d[1].cap_cc + z * sxy;
			for (uint64_t i = 0; i < sxy; i++) {
				if (lm1[c1[i]] != lm1[c2[i]]) { eq = 0; break; }                         /* :1160-1171 */
			}
		}
		*equal = eq;
	}
	for (int k = 0; k < 2; k++) {
		if (d[k].cap_lmap) for (uint64_t z = 0; z < sz; z++) free(d[k].cap_lmap[z]);
		free(d[k].cap_lmap); free(d[k].cap_N); free(d[k].cap_cc);
	}
	return rc;
}

/* operations::voxel_counts / centroids / bounding_boxes (src/operations.hpp:321-665), the per-pixel
 * loops of the reference over (component image, component -> label table) of every slice of the range.
 * The maps come back as arrays sorted by label (keys = labels as the unsigned type of the data width):
 * which 0: counts, 1 x uint64 per label; 1: centroids, 3 x double (integer sums, one division each);
 * 2: boxes, 6 x uint32 — every label of the stream's unique list has an entry, those absent from the
 * range keep the initial {max, max, max, 0, 0, 0} (:561-567). */
typedef struct { uint64_t label; uint64_t n, sx_, sy_; uint32_t x0, y0, x1, y1; uint32_t z; int seen; } lstat_t;
static int lstat_cmp(const void* a, const void* b) {
	const lstat_t* p = (const lstat_t*)a; const lstat_t* q = (const lstat_t*)b;
	return p->label < q->label ? -1 : (p->label > q->label ? 1 : 0);
}
int ckl_oracle_label_stats(
	const unsigned char* buf, uint64_t n, int which, int64_t z_start, int64_t z_end, uint64_t parallel,
	uint64_t** labels_out, void** values_out, uint64_t* n_out
) {
	header_t h;
	*labels_out = NULL; *values_out = NULL; *n_out = 0;
	if (n < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream.");
	if (header_read(&h, buf, n)) return 1;
	int64_t zs, szr;
	if (get_szr(&h, z_start, z_end, &zs, &szr)) return 1;
	const uint64_t sxy = (uint64_t)h.sx * h.sy;
	if (sxy * (uint64_t)szr == 0) {      /* :333-335: an empty map */
		*labels_out = (uint64_t*)xmalloc(8); *values_out = xmalloc(8);
		return 0;
	}
	const uint64_t lmask = h.data_width >= 8 ? ~0ull : ((1ull << (8 * h.data_width)) - 1ull);
	dec_ctx_t d;
	memset(&d, 0, sizeof d);
	d.cap_cc = (uint32_t*)xmalloc(sxy * (uint64_t)szr * 4);
	d.cap_N = (uint64_t*)xcalloc((size_t)szr, sizeof(uint64_t));
	d.cap_lmap = (uint64_t**)xcalloc((size_t)szr, sizeof(uint64_t*));
	int rc = dec_run(&d, buf, n, NULL, z_start < 0 ? 0 : z_start, z_end, parallel, 0, 0);
	lstat_t* rows = NULL;
	uint64_t nrows = 0;
	if (!rc) {
		uint64_t total = 0;
		for (int64_t zi = 0; zi < szr; zi++) total += d.cap_N[zi];
		/* boxes: one row per label of the unique list first (labels::unique, src/labels.hpp:393-422) */
		uint64_t n_uniq = 0, uniq_off = 0;
		const unsigned char* lb = buf + header_bytes(&h) + grid_index_bytes(&h);
		if (which == 2) {
			const uint64_t sw = (uint64_t)h.stored_data_width;
			if (h.label_format == FLAT) { n_uniq = rd(lb, 0, 8); uniq_off = 8; }
			else { n_uniq = rd(lb, sw, 8); uniq_off = sw + 8; }
		}
		rows = (lstat_t*)xcalloc((size_t)(total + n_uniq + 1), sizeof(lstat_t));
		for (uint64_t u = 0; u < n_uniq; u++) {
			lstat_t* r = &rows[nrows++];
			r->label = rd(lb, uniq_off + u * (uint64_t)h.stored_data_width, h.stored_data_width);
			r->x0 = r->y0 = 0xFFFFFFFFu; r->seen = 0;
		}
		for (int64_t zi = 0; zi < szr; zi++) {
			const uint64_t N = d.cap_N[zi];
			lstat_t* sub = (lstat_t*)xcalloc((size_t)N + 1, sizeof(lstat_t));
			for (uint64_t c = 0; c < N; c++) { sub[c].x0 = sub[c].y0 = 0xFFFFFFFFu; }
			const uint32_t* cc = d.cap_cc + (uint64_t)zi * sxy;
			for (uint64_t y = 0; y < h.sy; y++) {
				for (uint64_t x = 0; x < h.sx; x++) {
					lstat_t* r = &sub[cc[x + (uint64_t)h.sx * y]];
					r->n++; r->sx_ += x; r->sy_ += y;
					if ((uint32_t)x < r->x0) r->x0 = (uint32_t)x;
					if ((uint32_t)y < r->y0) r->y0 = (uint32_t)y;
					if ((uint32_t)x > r->x1) r->x1 = (uint32_t)x;
					if ((uint32_t)y > r->y1) r->y1 = (uint32_t)y;
				}
			}
			for (uint64_t c = 0; c < N; c++) {
				lstat_t* r = &rows[nrows++];
				*r = sub[c];
				r->label = d.cap_lmap[zi][c] & lmask;
				r->z = (uint32_t)(zs + zi);
				r->seen = 1;
			}
			free(sub);
		}
		qsort(rows, (size_t)nrows, sizeof(lstat_t), lstat_cmp);
		uint64_t nl = 0;
		for (uint64_t i = 0; i < nrows; i++) if (i == 0 || rows[i].label != rows[i - 1].label) nl++;
		uint64_t* labels = (uint64_t*)xmalloc((nl + 1) * 8);
		const size_t vb = which == 0 ? 8 : (which == 1 ? 24 : 24);
		unsigned char* values = (unsigned char*)xcalloc((size_t)nl + 1, vb);
		uint64_t k = 0;
		for (uint64_t i = 0; i < nrows;) {
			uint64_t j = i, cnt = 0, sumx = 0, sumy = 0, sumz = 0;
			uint32_t bx[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0 };
			/* a label outside the unique list (the background colour of a pin stream) is default-constructed
			 * by bbxes[label_map[i]] (:596): its minima start at 0 */
			int listed = 0;
			for (uint64_t q = i; q < nrows && rows[q].label == rows[i].label; q++) listed |= !rows[q].seen;
			if (which == 2 && !listed) bx[0] = bx[1] = bx[2] = 0;
			for (; j < nrows && rows[j].label == rows[i].label; j++) {
				const lstat_t* r = &rows[j];
				if (!r->seen) continue;
				cnt += r->n; sumx += r->sx_; sumy += r->sy_; sumz += r->n * (uint64_t)r->z;
				if (r->x0 < bx[0]) bx[0] = r->x0;
				if (r->y0 < bx[1]) bx[1] = r->y0;
				if (r->z < bx[2]) bx[2] = r->z;
				if (r->x1 > bx[3]) bx[3] = r->x1;
				if (r->y1 > bx[4]) bx[4] = r->y1;
				if (r->z > bx[5]) bx[5] = r->z;
			}
			labels[k] = rows[i].label;
			if (which == 0) ((uint64_t*)values)[k] = cnt;
			else if (which == 1) {
				double* v = (double*)values + 3 * k;
				v[0] = (double)sumx / (double)cnt; v[1] = (double)sumy / (double)cnt; v[2] = (double)sumz / (double)cnt;
			}
			else memcpy(values + 24 * k, bx, 24);
			k++;
			i = j;
		}
		*labels_out = labels; *values_out = values; *n_out = nl;
	}
	free(rows);
	for (int64_t zi = 0; zi < szr; zi++) if (d.cap_lmap) free(d.cap_lmap[zi]);
	free(d.cap_lmap); free(d.cap_N); free(d.cap_cc);
	return rc;
}

/* operations::mode_pooling_2x2x1 (src/operations.hpp:1201-1340): *out holds the per-slice streams one
 * after the other (release with ckl_oracle_free), lens_out their lengths (room for sz entries) */
int ckl_oracle_mode_pooling(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end, uint64_t parallel,
	unsigned char** out, uint64_t* out_len, uint64_t* lens_out, uint64_t* count
) {
	header_t h;
	*out = NULL; *out_len = 0; *count = 0;
	if (n < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream.");
	if (header_read(&h, buf, n)) return 1;
	int64_t zs, szr;
	if (get_szr(&h, z_start, z_end, &zs, &szr)) return 1;
	const int64_t ze = zs + szr;
	const uint64_t sx = h.sx, sy = h.sy;
	if (sx * sy == 0) return 0;      /* voxels == 0: no streams (:1213-1215) */
	const int w = h.data_width;
	/* decoded in the stream's own order; VOX() below addresses voxel (x, y) of slice zi in either */
	unsigned char* vol = (unsigned char*)xmalloc(sx * sy * (uint64_t)szr * (uint64_t)w);
	int rc = ckl_oracle_decompress(buf, n, vol, zs, ze, parallel, 0, 0);
	if (rc) { free(vol); return rc; }
	const uint64_t osx = (sx + 1) >> 1, osy = (sy + 1) >> 1, osxy = osx * osy;
	unsigned char* oimg = (unsigned char*)xmalloc(osxy * (uint64_t)w);
	unsigned char* all = NULL;
	uint64_t total = 0;
	for (int64_t zi = 0; zi < szr && !rc; zi++) {
		for (uint64_t oy = 0; oy < osy; oy++) {
			for (uint64_t ox = 0; ox < osx; ox++) {
				const uint64_t x = 2 * ox, y = 2 * oy;
#define VOX(X, Y) load_label(vol, w, h.fortran_order ? ((X) + sx * ((Y) + sy * (uint64_t)zi)) : ((uint64_t)zi + (uint64_t)szr * ((Y) + sy * (X))))
				uint64_t v = VOX(x, y);
				if (x + 1 < sx && y + 1 < sy) {
					const uint64_t a = v, b = VOX(x + 1, y), c = VOX(x, y + 1), dd = VOX(x + 1, y + 1);
					v = (a == b) ? a : (a == c) ? a : (b == c) ? b : dd;                  /* :1262-1273 */
				}
#undef VOX
				store_out(oimg, w, ox + osx * oy, v);
			}
		}
		unsigned char* one = NULL; uint64_t len = 0;
		rc = ckl_oracle_compress(oimg, w, 0, (int64_t)osx, (int64_t)osy, 1, 0, 1, 0, 0, 1, 0, 1, &one, &len);      /* :1294-1297 */
		if (!rc) {
			all = (unsigned char*)xrealloc(all, total + len + 1);
			memcpy(all + total, one, len);
			total += len;
			lens_out[zi] = len;
			free(one);
		}
	}
	free(oimg); free(vol);
	if (rc) { free(all); return rc; }
	*out = all ? all : (unsigned char*)xmalloc(1);
	*out_len = total; *count = (uint64_t)szr;
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * point clouds: dual_graph::extract_contours (src/dual_graph.hpp:133-275) + operations::point_cloud
 * (src/operations.hpp:183-262)
 * ------------------------------------------------------------------------------------------ */
enum { DG_RIGHT = 1, DG_LEFT = 2, DG_DOWN = 4, DG_UP = 8, DG_VISITED = 16 };

/* compute_next_move (src/dual_graph.hpp:66-131): turn order relative to the last move */
static uint8_t dg_next_move(int clockwise, uint8_t last, uint8_t allowed) {
	static const uint8_t cw[4][4] = {
		{ DG_DOWN, DG_RIGHT, DG_UP, DG_LEFT },      /* last = RIGHT */
		{ DG_UP, DG_LEFT, DG_DOWN, DG_RIGHT },      /* last = LEFT */
		{ DG_RIGHT, DG_UP, DG_LEFT, DG_DOWN },      /* last = UP */
		{ DG_LEFT, DG_DOWN, DG_RIGHT, DG_UP },      /* otherwise (DOWN) */
	};
	static const uint8_t ccw[4][4] = {
		{ DG_UP, DG_RIGHT, DG_DOWN, DG_LEFT },
		{ DG_DOWN, DG_LEFT, DG_UP, DG_RIGHT },
		{ DG_LEFT, DG_UP, DG_RIGHT, DG_DOWN },
		{ DG_RIGHT, DG_DOWN, DG_LEFT, DG_UP },
	};
	const int row = last == DG_RIGHT ? 0 : last == DG_LEFT ? 1 : last == DG_UP ? 2 : 3;
	const uint8_t* pref = clockwise ? cw[row] : ccw[row];
	for (int k = 0; k < 4; k++) if (allowed & pref[k]) return pref[k];
	return 0;
}

typedef struct { uint32_t* v; uint64_t n, cap; } u32vec_t;
static void u32vec_push(u32vec_t* a, uint32_t x) {
	if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 64; a->v = (uint32_t*)xrealloc(a->v, a->cap * 4); }
	a->v[a->n++] = x;
}

/* extract_contours_helper + merge_contours_via_vcg_coloring: merged[c] receives the contour nodes
 * of component c (vcg is modified: border bits cleared, visited bits set) */
static void dg_extract_contours(uint8_t* vcg, const uint32_t* cc, uint64_t N, int64_t sx, int64_t sy, u32vec_t* merged) {
	for (int64_t i = 0; i < sx; i++) {                                                 /* :139-143 */
		vcg[i] &= (uint8_t)~DG_UP;
		vcg[i + sx * (sy - 1)] &= (uint8_t)~DG_DOWN;
	}
	for (int64_t i = 0; i < sy; i++) {
		vcg[sx * i] &= (uint8_t)~DG_LEFT;
		vcg[sx - 1 + sx * i] &= (uint8_t)~DG_RIGHT;
	}
	int64_t move_amt[9] = { 0 };
	move_amt[DG_RIGHT] = 1; move_amt[DG_LEFT] = -1; move_amt[DG_DOWN] = sx; move_amt[DG_UP] = -sx;
	u32vec_t cur = { 0 };
	int64_t start = 0, y = 0;
	(void)N;
	for (;;) {
		/* VCGGraph::next_contour (:40-61) */
		int found = 0;
		int64_t x = start - sx * y;
		for (; y < sy; y++) {
			for (; x < sx; x++, start++) {
				if ((vcg[start] & 0x33) < 3 || (x < sx - 1 && (vcg[start + 1] & 0xF2) == 0)) { found = 1; break; }
			}
			if (found) break;
			x = 0;
		}
		if (!found) break;
		cur.n = 0;
		int64_t node = start;
		uint8_t allowed = vcg[node] & 15;
		uint64_t already = (vcg[node] >> 4) > 0;
		if (allowed == 0) {
			vcg[node] |= DG_VISITED;
			u32vec_push(&cur, (uint32_t)node);
		}
		else {
			u32vec_push(&cur, (uint32_t)start);
			const int clockwise = ((vcg[start] & 1) == 0) || (vcg[start] == 0x1C);       /* :177 */
			const uint8_t ending = dg_next_move(clockwise, DG_UP, allowed);
			uint8_t next = ending;
			do {
				node += move_amt[next];
				u32vec_push(&cur, (uint32_t)node);
				already += (vcg[node] >> 4) > 0;
				vcg[node] |= DG_VISITED;
				allowed = vcg[node] & 15;
				next = dg_next_move(clockwise, next, allowed);
			} while (!(node == start && next == ending));
		}
		start++;
		if (cur.n == 0 || cur.n == already) continue;
		/* rotate: the smallest node first (:203-211), then merge by component (:223-241) */
		uint64_t at = 0;
		for (uint64_t i = 1; i < cur.n; i++) if (cur.v[i] < cur.v[at]) at = i;
		u32vec_t* m = &merged[cc[cur.v[at]]];
		const int front = m->n > 0 && m->v[0] > cur.v[at];
		const uint64_t old = m->n;
		while (m->cap < old + cur.n) { m->cap = m->cap ? 2 * m->cap : 64; m->v = (uint32_t*)xrealloc(m->v, m->cap * 4); }
		if (front) memmove(m->v + cur.n, m->v, old * 4);
		uint32_t* dst = front ? m->v : m->v + old;
		memcpy(dst, cur.v + at, (cur.n - at) * 4);
		memcpy(dst + (cur.n - at), cur.v, at * 4);
		m->n = old + cur.n;
	}
	free(cur.v);
}

static int cmp_u64(const void* a, const void* b) {
	const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

/* operations::point_cloud with parallel = 1 (slices in order): the labels that own points, ascending,
 * and for each the (x, y, z) uint16 triples in the order the reference appends them.
 * labels_out[n_out], offsets_out[n_out + 1] (in points), points_out[3 * offsets_out[n_out]];
 * release each with ckl_oracle_free. */
int ckl_oracle_point_cloud(
	const unsigned char* buf, uint64_t n, int64_t z_start, int64_t z_end,
	const uint64_t* labels, uint64_t n_labels, int has_labels, int skip_background,
	uint64_t** labels_out, uint64_t** offsets_out, uint16_t** points_out, uint64_t* n_out
) {
	header_t h;
	*labels_out = NULL; *offsets_out = NULL; *points_out = NULL; *n_out = 0;
	if (n < HEADER_BYTES_V0) FAIL("crackle: Input too small to be a valid stream.");
	if (header_read(&h, buf, n)) return 1;
	int64_t zs, szr;
	if (get_szr(&h, z_start, z_end, &zs, &szr)) return 1;
	const uint64_t sx = h.sx, sy = h.sy, sxy = sx * sy;
	if (sxy == 0) return 0;
	dec_ctx_t d;
	memset(&d, 0, sizeof d);
	d.cap_cc = (uint32_t*)xmalloc(sxy * (uint64_t)szr * 4);
	d.cap_N = (uint64_t*)xcalloc((size_t)szr, sizeof(uint64_t));
	d.cap_lmap = (uint64_t**)xcalloc((size_t)szr, sizeof(uint64_t*));
	int rc = dec_run(&d, buf, n, NULL, zs, zs + szr, 1, 0, 0);
	uint8_t* vcg = (uint8_t*)xmalloc(sxy);
	/* per slice and component: the merged contour */
	u32vec_t** per = (u32vec_t**)xcalloc((size_t)szr, sizeof(u32vec_t*));
	for (int64_t zi = 0; zi < szr && !rc; zi++) {
		rc = ckl_oracle_slice_vcg(buf, n, zs + zi, vcg);
		if (rc) break;
		per[zi] = (u32vec_t*)xcalloc((size_t)d.cap_N[zi] + 1, sizeof(u32vec_t));
		dg_extract_contours(vcg, d.cap_cc + (uint64_t)zi * sxy, d.cap_N[zi], (int64_t)sx, (int64_t)sy, per[zi]);
	}
	if (!rc) {
		/* the labels with points, ascending */
		uint64_t total_comp = 0;
		for (int64_t zi = 0; zi < szr; zi++) total_comp += d.cap_N[zi];
		uint64_t* keys = (uint64_t*)xmalloc((total_comp + 1) * 8);
		uint64_t nk = 0;
		/* point_cloud<LABEL> holds the labels as the unsigned type of the data width (:274-300) */
		const uint64_t lmask = h.data_width >= 8 ? ~0ull : ((1ull << (8 * h.data_width)) - 1);
		for (int64_t zi = 0; zi < szr; zi++) {
			for (uint64_t c = 0; c < d.cap_N[zi]; c++) {
				const uint64_t L = d.cap_lmap[zi][c] & lmask;
				int sel = 1;
				if (skip_background && L == 0) sel = 0;
				if (sel && has_labels) {
					sel = 0;
					for (uint64_t k = 0; k < n_labels; k++) if (labels[k] == L) { sel = 1; break; }
				}
				/* ptc[current_label] creates the entry even when the component has no contour nodes */
				if (sel) keys[nk++] = L;
			}
		}
		qsort(keys, nk, 8, cmp_u64);
		uint64_t nu = 0;
		for (uint64_t i = 0; i < nk; i++) if (i == 0 || keys[i] != keys[i - 1]) keys[nu++] = keys[i];
		uint64_t* off = (uint64_t*)xcalloc(nu + 2, 8);
		for (int pass = 0; pass < 2; pass++) {
			uint64_t* fill = NULL;
			uint16_t* pts = NULL;
			if (pass == 1) {
				uint64_t run = 0;
				for (uint64_t i = 0; i <= nu; i++) { const uint64_t c = off[i]; off[i] = run; run += c; }
				fill = (uint64_t*)xmalloc((nu + 1) * 8);
				memcpy(fill, off, (nu + 1) * 8);
				pts = (uint16_t*)xmalloc((off[nu] * 3 + 1) * 2);
				*points_out = pts;
			}
			for (int64_t zi = 0; zi < szr; zi++) {
				for (uint64_t c = 0; c < d.cap_N[zi]; c++) {
					const uint64_t L = d.cap_lmap[zi][c] & lmask;
					if (skip_background && L == 0) continue;
					uint64_t lo = 0, hi = nu;
					while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (keys[mid] < L) lo = mid + 1; else hi = mid; }
					if (lo == nu || keys[lo] != L) continue;      /* not selected */
					const u32vec_t* m = &per[zi][c];
					if (pass == 0) { off[lo] += m->n; continue; }
					for (uint64_t k = 0; k < m->n; k++) {
						const uint32_t loc = m->v[k];
						const uint16_t yy = (uint16_t)(loc / h.sx);                          /* :246-247: 16-bit truncation */
						const uint16_t xx = (uint16_t)(loc - h.sx * yy);
						uint16_t* q = pts + 3 * fill[lo]++;
						q[0] = xx; q[1] = yy; q[2] = (uint16_t)(zs + zi);
					}
				}
			}
			free(fill);
		}
		*labels_out = keys; *offsets_out = off; *n_out = nu;
	}
	for (int64_t zi = 0; zi < szr; zi++) {
		if (per[zi]) { for (uint64_t c = 0; c <= d.cap_N[zi]; c++) free(per[zi][c].v); free(per[zi]); }
		if (d.cap_lmap) free(d.cap_lmap[zi]);
	}
	free(per); free(vcg); free(d.cap_lmap); free(d.cap_N); free(d.cap_cc);
	return rc;
}

int ckl_oracle_slice_vcg(const unsigned char* buf, uint64_t n, int64_t z, uint8_t* vcg_out) {
	dec_ctx_t d;
	memset(&d, 0, sizeof d);
	if (header_read(&d.head, buf, n)) return 1;
	const header_t* h = &d.head;
	if (z < 0 || z >= (int64_t)h->sz) FAIL("crackle: z out of range");
	d.buf = buf; d.n = n;
	d.labels_binary = buf + header_bytes(h) + grid_index_bytes(h);
	d.order = h->markov_model_order;
	if (d.order > 0) d.model = mk_from_stored(d.labels_binary + h->num_label_bytes, markov_model_bytes(h), d.order);
	int rc = dec_z_index(&d);
	if (!rc) {
		rc = slice_to_vcg(buf + d.z_index[z], d.z_index[z + 1] - d.z_index[z], h->sx, h->sy, h->crack_format == PERMISSIBLE, d.model, d.order, vcg_out);
		if (rc) snprintf(g_err, sizeof g_err, "crackle: malformed crack code");
	}
	dec_free(&d, 0);
	return rc;
}

/* operations::voxel_connectivity_graph (src/operations.hpp:667-826), whole volume: the four
 * in-plane bits of every voxel are the slice's voxel connectivity graph as the crack decoder
 * leaves it (crack_code_to_vcg; bit0 +x, bit1 -x, bit2 +y, bit3 -y), connectivity 6 adds bit4
 * (+z) / bit5 (-z) between voxels of equal label in neighbouring slices and marks the first
 * slice's -z and the last slice's +z as passable.  vcg: sx*sy*sz bytes, x fastest. */
int ckl_oracle_voxel_connectivity_graph(
	const unsigned char* buf, uint64_t n, int connectivity, uint64_t parallel, uint8_t* vcg
) {
	if (connectivity != 4 && connectivity != 6) FAIL("crackle: voxel_connectivity_graph: only connectivity 4 and 6 are currently supported.");
	header_t h;
	if (header_read(&h, buf, n)) return 1;
	const uint64_t sxy = (uint64_t)h.sx * h.sy;
	if (sxy * h.sz == 0) return 0;
	for (uint64_t z = 0; z < h.sz; z++) {
		if (ckl_oracle_slice_vcg(buf, n, (int64_t)z, vcg + z * sxy)) return 1;
	}
	if (h.sz == 1 || connectivity == 4) return 0;     /* :757-759 */
	/* the reference compares label_map[ccl] of neighbouring slices: the decoded labels */
	uint8_t* lab = (uint8_t*)xmalloc(sxy * h.sz * (uint64_t)h.data_width);
	if (ckl_oracle_decompress(buf, n, lab, 0, -1, parallel, 0, 0)) { free(lab); return 1; }
	const int w = h.data_width;
	for (uint64_t z = 1; z < h.sz; z++) {
		for (uint64_t y = 0; y < h.sy; y++) {
			for (uint64_t x = 0; x < h.sx; x++) {
				const uint64_t loc = x + (uint64_t)h.sx * y;
				/* decompress honours the stream's memory order */
				const uint64_t top = h.fortran_order ? loc + (z - 1) * sxy : (z - 1) + (uint64_t)h.sz * (y + (uint64_t)h.sy * x);
				const uint64_t bot = h.fortran_order ? loc + z * sxy : z + (uint64_t)h.sz * (y + (uint64_t)h.sy * x);
				if (memcmp(lab + top * w, lab + bot * w, (size_t)w) == 0) {
					vcg[loc + (z - 1) * sxy] |= 0x10;
					vcg[loc + z * sxy] |= 0x20;
				}
			}
		}
	}
	for (uint64_t loc = 0; loc < sxy; loc++) {        /* :812-821 */
		vcg[loc] |= 0x20;
		vcg[loc + (h.sz - 1) * sxy] |= 0x10;
	}
	free(lab);
	return 0;
}

/* reencode_with_markov_order (src/crackle.hpp:858-984): every slice's crack code is taken
 * apart into chains of symbols (crack_code_to_symbols, :394-411), turned back into code points
 * (symbols_to_codepoints, src/crackcodes.hpp:128-183) and packed again under the new order
 * (pack_codepoints / markov::gather_statistics + compress); header, z-index and model are
 * rewritten, label section and the trailing crcs are copied.  Version-0 streams (no crcs)
 * are not handled here. */
int ckl_oracle_reencode(
	const unsigned char* buf, uint64_t n, int markov_order, uint64_t parallel, unsigned char** out, uint64_t* out_len
) {
	(void)parallel;
	dec_ctx_t d;
	memset(&d, 0, sizeof d);
	if (n < HEADER_BYTES) FAIL("crackle: Input too small to be a valid stream. Bytes: %llu", (unsigned long long)n);
	if (header_read(&d.head, buf, n)) return 1;
	header_t* h = &d.head;
	if (h->format_version == 0) FAIL("crackle oracle: reencode of version 0 streams is not restated");
	if (markov_order < 0 || markov_order > 15) FAIL("crackle oracle: markov order out of range");
	if (h->markov_model_order == markov_order) {   /* :887-889 */
		*out = (unsigned char*)xmalloc(n ? n : 1);
		memcpy(*out, buf, n);
		*out_len = n;
		return 0;
	}
	d.buf = buf; d.n = n;
	if (header_bytes(h) + grid_index_bytes(h) + h->num_label_bytes + markov_model_bytes(h) + 4 * ((uint64_t)h->sz + 1) > n)
		FAIL("crackle: get_crack_code_offsets: Unable to read past end of buffer.");
	d.labels_binary = buf + header_bytes(h) + grid_index_bytes(h);
	d.order = h->markov_model_order;
	if (d.order > 0) d.model = mk_from_stored(d.labels_binary + h->num_label_bytes, markov_model_bytes(h), d.order);
	if (dec_z_index(&d)) { dec_free(&d, 0); return 1; }
	const uint64_t sz = h->sz;
	const uint64_t tail = d.z_index[sz];        /* labels crc + slice crcs follow the codes */
	if (tail + 4 * (sz + 1) > n) { dec_free(&d, 0); FAIL("crackle: Unable to read past end of buffer."); }

	chainset_t* sets = (chainset_t*)xcalloc(sz ? sz : 1, sizeof(chainset_t));
	int rc = 0;
	for (uint64_t z = 0; z < sz && !rc; z++) {
		const unsigned char* code = buf + d.z_index[z];
		const uint64_t code_len = d.z_index[z + 1] - d.z_index[z];
		size_t n_nodes; int bad;
		uint64_t* nodes = read_boc_index(code, code_len, h->sx, h->sy, &n_nodes, &bad);
		const uint64_t index_size = code_len >= 4 ? 4 + rd(code, 0, 4) : 0;
		if (bad || index_size > code_len || code_len < 4) { free(nodes); rc = 1; break; }
		uint8_t* moves; size_t n_moves;
		if (d.order == 0) {
			n_moves = (size_t)(code_len - index_size) * 4;
			moves = (uint8_t*)xmalloc(n_moves + 1);
			uint8_t last = 0;
			size_t m = 0;
			for (uint64_t i = index_size; i < code_len; i++) {
				for (int j = 0; j < 4; j++) {
					uint8_t mv = (uint8_t)(((code[i] >> (2 * j)) & 3) + last) & 3;
					last = mv;
					moves[m++] = mv;
				}
			}
		}
		else moves = mk_decode(code + index_size, code_len - index_size, d.model, d.order, &n_moves);
		unsigned char* symbols = (unsigned char*)xmalloc(n_moves + 1);
		size_t nc;
		dchain_t* chains = moves_to_symbols(moves, n_moves, nodes, n_nodes, symbols, &nc);
		chainset_t* cs = &sets[z];
		cs->chains = (chain_t*)xcalloc(nc ? nc : 1, sizeof(chain_t));
		cs->n = cs->cap = nc;
		for (size_t c = 0; c < nc; c++) {
			const size_t len = chains[c].end - chains[c].begin;
			cs->chains[c].node = chains[c].node;
			cs->chains[c].codes = (uint8_t*)xmalloc(2 * len + 2);
			cs->chains[c].n = symbols_to_codepoints(symbols + chains[c].begin, len, cs->chains[c].codes);
		}
		/* the unordered map of chains is consumed in ascending node order (pack_codepoints :460-464) */
		qsort(cs->chains, cs->n, sizeof(chain_t), chain_cmp);
		free(chains); free(symbols); free(moves); free(nodes);
	}
	if (rc) {
		for (uint64_t z = 0; z < sz; z++) chainset_free(&sets[z]);
		free(sets); dec_free(&d, 0);
		FAIL("crackle: malformed crack code");
	}

	bytes_t* codes = (bytes_t*)xcalloc(sz ? sz : 1, sizeof(bytes_t));
	bytes_t stored = { 0 };
	if (markov_order > 0) {
		const size_t rows = (size_t)1 << (2 * markov_order);
		uint32_t* stats = (uint32_t*)xcalloc(rows * 4, sizeof(uint32_t));
		for (uint64_t z = 0; z < sz; z++) {
			size_t nd;
			uint8_t* dd = slice_diffcodes(&sets[z], &nd);
			mk_stats_slice(dd, nd, markov_order, stats);
			free(dd);
		}
		uint8_t* model = (uint8_t*)xmalloc(rows * 4);
		mk_stats_to_model(stats, rows, model);
		mk_to_stored(model, rows, &stored);
		for (uint64_t z = 0; z < sz; z++) {
			size_t nd;
			uint8_t* dd = slice_diffcodes(&sets[z], &nd);
			write_boc_index(&sets[z], h->sx, h->sy, &codes[z]);
			mk_encode(dd, nd, model, markov_order, &codes[z]);
			free(dd);
		}
		free(model); free(stats);
	}
	else {
		for (uint64_t z = 0; z < sz; z++) pack_codepoints(&sets[z], h->sx, h->sy, &codes[z]);
	}

	bytes_t o = { 0 };
	header_t nh = *h;
	nh.markov_model_order = markov_order;
	header_write(&nh, &o);
	const size_t zi0 = o.n;
	for (uint64_t z = 0; z < sz; z++) bput(&o, codes[z].n, 4);
	bput(&o, ckl_oracle_crc32c(o.p + zi0, 4 * sz), 4);
	bpush(&o, d.labels_binary, h->num_label_bytes);
	if (stored.n) bpush(&o, stored.p, stored.n);
	for (uint64_t z = 0; z < sz; z++) if (codes[z].n) bpush(&o, codes[z].p, codes[z].n);
	bpush(&o, buf + tail, 4 * (sz + 1));
	for (uint64_t z = 0; z < sz; z++) { chainset_free(&sets[z]); free(codes[z].p); }
	free(sets); free(codes); free(stored.p);
	dec_free(&d, 0);
	*out = o.p;
	*out_len = o.n;
	return 0;
}

/* lib::max_label / pixel_pairs (src/lib.hpp:224-256) of one slab plus its first and last
 * voxel: what a sharded encoder all-gathers. */
int ckl_oracle_stats(
	const void* labels_v, int dtype_bytes, int64_t sx, int64_t sy, int64_t sz,
	uint64_t* max_label, uint64_t* pixel_pairs, uint64_t* first_voxel, uint64_t* last_voxel
) {
	const int64_t voxels = sx * sy * sz;
	uint64_t* labels = widen(labels_v, dtype_bytes, (uint64_t)voxels);
	uint64_t mx = 0, pairs = 0;
	for (int64_t i = 0; i < voxels; i++) if (labels[i] > mx) mx = labels[i];
	for (int64_t i = 1; i < voxels; i++) pairs += (labels[i] == labels[i - 1]);
	*max_label = mx; *pixel_pairs = pairs;
	*first_voxel = voxels ? labels[0] : 0;
	*last_voxel = voxels ? labels[voxels - 1] : 0;
	free(labels);
	return 0;
}

/* markov::gather_statistics (src/markov.hpp:193-220) of one slab under a given crack format */
int ckl_oracle_markov_hist(
	const void* labels_v, int dtype_bytes, int64_t sx, int64_t sy, int64_t sz,
	int crack_format, uint64_t order, uint32_t* hist
) {
	if (order == 0 || order > 15) FAIL("ckl_oracle: bad markov order");
	const int64_t voxels = sx * sy * sz;
	const size_t rows = (size_t)1 << (2 * order);
	memset(hist, 0, rows * 4 * sizeof(uint32_t));
	if (voxels == 0) return 0;
	uint64_t* labels = widen(labels_v, dtype_bytes, (uint64_t)voxels);
	for (int64_t z = 0; z < sz; z++) {
		chainset_t cs;
		create_crack_codes(labels + sx * sy * z, sx, sy, crack_format == PERMISSIBLE, &cs);
		size_t n;
		uint8_t* d = slice_diffcodes(&cs, &n);
		mk_stats_slice(d, n, (int)order, hist);
		free(d);
		chainset_free(&cs);
	}
	free(labels);
	return 0;
}

/* ------------------------------------------------------------------ */
/* pins (src/pins.hpp:95-403, src/labels.hpp:157-344)                   */
/* ------------------------------------------------------------------ */
#include "ckl_oracle_pins.inc"
