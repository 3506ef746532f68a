// host_inflate.cpp — raw DEFLATE (RFC 1951) decoding of one whole buffer into one whole buffer, and CRC-32 (RFC 1952) —
// what host::BgzfStream does to every 64 KiB block of a BAM or a BGZF FASTQ (the reference reads both through gunzip /
// htslib: src/common.rs:83-157).  zlib's inflate() is a streaming decoder that can stop and resume after any byte; a
// BGZF block is never resumed, and with the reader's threads behind a CPU quota (the GPU boxes grant 16 CPUs of their
// 256) `sam statistics` on a 20 M-record file was 14 CPU-seconds of inflate() + crc32() for 1.2 s of wall time.  This
// decoder keeps 64 bits of input in a register, resolves a symbol with one table lookup (11 bits at once for literals and
// lengths, 8 for distances, second-level tables behind them for the rare longer codes), decodes up to three literals per
// refill, and copies matches eight bytes at a time; the CRC folds 64 bytes per step with carry-less multiplies.
// Anything irregular — a code that is over-subscribed or incomplete, a distance before the start of the output, input or
// output running out — makes it give up, and the caller hands the block to zlib, whose verdict then stands.
#include "host_common.h"

#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace host {

namespace {

constexpr int kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr int kMaxLitLen = 15;
// table entry: bits 0-7 = bits the codeword takes (in a second-level table: the bits BEHIND the first kLitBits / kDistBits),
// bits 8-11 = extra bits, bits 12-15 = kind, bits 16-31 = literal / base value / start of the second-level table
constexpr uint32_t kLiteral = 1u << 12, kEob = 2u << 12, kSub = 4u << 12, kBad = 8u << 12;

constexpr uint32_t kLiteral2 = 1u << 8;         // (pairs only, in Tables::pair) two literals: bits 16-23 the first, 24-31 the second
struct Tables {
	uint32_t pair[1 << kLitBits];               // lit[0 .. 2^kLitBits) again, with two literals per entry where both codes fit the index
	uint32_t lit[(1 << kLitBits) + 1024];       // 288 symbols: the second-level tables of all prefixes stay well below this
	uint32_t dist[(1 << kDistBits) + 512];
	uint32_t pre[1 << kPreBits];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len)
{
	uint32_t r = 0;
	for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
	return r;
}

// kind: 0 = literal/length alphabet, 1 = distance alphabet, 2 = code-length alphabet.  false = not a complete prefix code
// (the one exception DEFLATE allows — a single distance code of one bit — included), or the tables would not fit.
bool build(const uint8_t *lens, int nsym, int kind, uint32_t *table, int table_cap, int primary_bits)
{
	int count[kMaxLitLen + 1] = {0};
	for (int s = 0; s < nsym; s++) count[lens[s]]++;
	if (count[0] == nsym) {
		// no code at all: legal for the distance alphabet of a block of literals only; every lookup is an error
		if (kind != 1) return false;
		for (int i = 0; i < (1 << primary_bits); i++) table[i] = kBad | 1u;
		return true;
	}
	int left = 1, maxlen = 0;
	for (int l = 1; l <= kMaxLitLen; l++) {
		left = (left << 1) - count[l];
		if (left < 0) return false;                                   // over-subscribed
		if (count[l]) maxlen = l;
	}
	const bool single = kind == 1 && count[1] == 1 && count[0] == nsym - 1;   // one distance code of length 1: its other half stays invalid
	if (left != 0 && !single) return false;                           // incomplete
	uint32_t next_code[kMaxLitLen + 2];
	{
		uint32_t code = 0;
		count[0] = 0;                                                 // unused symbols take no code
		for (int l = 1; l <= kMaxLitLen; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
	}
	const int psize = 1 << primary_bits;
	for (int i = 0; i < psize; i++) table[i] = kBad | 1u;
	// second-level tables: one per primary-bits prefix of the codes longer than primary_bits, sized by the longest code under it
	int sub_bits[1 << kLitBits];
	if (maxlen > primary_bits) {
		memset(sub_bits, 0, sizeof(int) * (size_t)psize);
		uint32_t nc[kMaxLitLen + 2];
		memcpy(nc, next_code, sizeof nc);
		for (int s = 0; s < nsym; s++) {
			const int l = lens[s];
			if (l <= primary_bits) { if (l) nc[l]++; continue; }
			const uint32_t rev = reverse_bits(nc[l]++, l);
			const int p = (int)(rev & (uint32_t)(psize - 1));
			if (l - primary_bits > sub_bits[p]) sub_bits[p] = l - primary_bits;
		}
		int at = psize;
		for (int p = 0; p < psize; p++) {
			if (!sub_bits[p]) continue;
			if (at + (1 << sub_bits[p]) > table_cap) return false;
			table[p] = kSub | ((uint32_t)at << 16) | (uint32_t)sub_bits[p];
			for (int i = 0; i < (1 << sub_bits[p]); i++) table[at + i] = kBad | 1u;
			at += 1 << sub_bits[p];
		}
	}
	for (int s = 0; s < nsym; s++) {
		const int l = lens[s];
		if (!l) continue;
		const uint32_t rev = reverse_bits(next_code[l]++, l);
		uint32_t e;
		if (kind == 0) {
			if (s < 256) e = kLiteral | ((uint32_t)s << 16);
			else if (s == 256) e = kEob;
			else if (s <= 285) e = ((uint32_t)kLenBase[s - 257] << 16) | ((uint32_t)kLenExtra[s - 257] << 8);
			else e = kBad;                                            // 286, 287: in the fixed code, never valid in data
		} else if (kind == 1) {
			e = s < 30 ? ((uint32_t)kDistBase[s] << 16) | ((uint32_t)kDistExtra[s] << 8) : kBad;
		} else {
			e = (uint32_t)s << 16;
		}
		if (l <= primary_bits) {
			e |= (uint32_t)l;
			for (uint32_t i = rev; i < (uint32_t)psize; i += 1u << l) table[i] = e;
		} else {
			const int p = (int)(rev & (uint32_t)(psize - 1));
			const uint32_t start = table[p] >> 16;
			const int sb = (int)(table[p] & 0xffu);
			e |= (uint32_t)(l - primary_bits);
			for (uint32_t i = rev >> primary_bits; i < (1u << sb); i += 1u << (l - primary_bits)) table[start + i] = e;
		}
	}
	return true;
}

// Literal-heavy streams (base qualities, packed bases) spend their time in a chain of dependent lookups, one per literal:
// index -> entry -> shift -> index.  Where the codes of two consecutive literals fit the 11 index bits together, one entry
// of `pair` yields both (the second is looked up here, once per table, instead of once per occurrence).
void make_pairs(Tables &t)
{
	for (uint32_t i = 0; i < (1u << kLitBits); i++) {
		const uint32_t e1 = t.lit[i];
		t.pair[i] = e1;
		if (!(e1 & kLiteral)) continue;
		const uint32_t l1 = e1 & 0xffu;
		if (l1 >= (uint32_t)kLitBits) continue;
		const uint32_t e2 = t.lit[i >> l1];
		if (!(e2 & kLiteral) || (e2 & 0xffu) > (uint32_t)kLitBits - l1) continue;
		t.pair[i] = kLiteral | kLiteral2 | (l1 + (e2 & 0xffu)) | (e1 & 0x00ff0000u) | ((e2 & 0x00ff0000u) << 8);
	}
}

struct Bits {
	const uint8_t *ip, *end;
	uint64_t buf = 0;
	int cnt = 0;
	size_t overrun = 0;                      // zero bytes imagined behind the end of the input (the last refills of a block reach
	                                         // past it; that is an error only once one of those bits has been consumed)
	inline bool past_end() const { return overrun * 8 > (size_t)cnt; }
	inline void refill()
	{
		if (end - ip >= 8) {
			uint64_t w;
			memcpy(&w, ip, 8);
			buf |= w << cnt;
			ip += (63 - cnt) >> 3;
			cnt |= 56;
		} else {
			while (cnt <= 56) {
				if (ip < end) buf |= (uint64_t)*ip++ << cnt;
				else overrun++;
				cnt += 8;
			}
		}
	}
	inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
	inline void drop(int n) { buf >>= n; cnt -= n; }
	inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

}  // namespace

bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
	static thread_local Tables t;
	static thread_local bool have_fixed = false;
	static thread_local Tables fixed;
	Bits b;
	b.ip = in; b.end = in + in_len;
	uint8_t *op = out, *const oend = out + out_len;
	for (;;) {
		b.refill();
		const uint32_t bfinal = b.take(1), btype = b.take(2);
		const Tables *tb = &t;
		if (btype == 0) {
			// stored: LEN / NLEN behind the next byte boundary
			b.drop(b.cnt & 7);
			b.refill();
			const uint32_t len = b.take(16), nlen = b.take(16);
			if ((len ^ nlen) != 0xffffu) return false;
			// the bytes still in the bit buffer come first
			uint32_t left = len;
			while (left && b.cnt >= 8) { if (op == oend) return false; *op++ = (uint8_t)b.take(8); left--; }
			if (b.past_end()) return false;
			if (left) {
				if ((size_t)(b.end - b.ip) < left || (size_t)(oend - op) < left) return false;
				memcpy(op, b.ip, left);
				b.ip += left; op += left;
				b.buf = 0; b.cnt = 0;
			}
		} else if (btype == 1 || btype == 2) {
			if (btype == 1) {
				if (!have_fixed) {
					uint8_t l[288 + 32];
					for (int i = 0; i < 144; i++) l[i] = 8;
					for (int i = 144; i < 256; i++) l[i] = 9;
					for (int i = 256; i < 280; i++) l[i] = 7;
					for (int i = 280; i < 288; i++) l[i] = 8;
					for (int i = 0; i < 32; i++) l[288 + i] = 5;
					if (!build(l, 288, 0, fixed.lit, (int)(sizeof fixed.lit / 4), kLitBits) || !build(l + 288, 32, 1, fixed.dist, (int)(sizeof fixed.dist / 4), kDistBits)) return false;
					make_pairs(fixed);
					have_fixed = true;
				}
				tb = &fixed;
			} else {
				const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
				if (hlit > 286 || hdist > 30) return false;
				static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
				uint8_t pl[19] = {0};
				b.refill();
				for (uint32_t i = 0; i < hclen; i++) {
					if (b.cnt < 3) b.refill();
					pl[order[i]] = (uint8_t)b.take(3);
				}
				if (!build(pl, 19, 2, t.pre, 1 << kPreBits, kPreBits)) return false;
				uint8_t l[286 + 30 + 140];
				uint32_t n = 0;
				const uint32_t total = hlit + hdist;
				while (n < total) {
					b.refill();
					const uint32_t e = t.pre[b.peek(kPreBits)];
					if (e & kBad) return false;
					b.drop((int)(e & 0xffu));
					const uint32_t sym = e >> 16;
					if (sym < 16) { l[n++] = (uint8_t)sym; continue; }
					uint32_t rep, val = 0;
					if (sym == 16) { if (n == 0) return false; val = l[n - 1]; rep = 3 + b.take(2); }
					else if (sym == 17) rep = 3 + b.take(3);
					else rep = 11 + b.take(7);
					if (n + rep > total) return false;
					memset(l + n, (int)val, rep);
					n += rep;
				}
				if (b.past_end()) return false;
				if (l[256] == 0) return false;                            // no end-of-block code
				if (!build(l, (int)hlit, 0, t.lit, (int)(sizeof t.lit / 4), kLitBits) || !build(l + hlit, (int)hdist, 1, t.dist, (int)(sizeof t.dist / 4), kDistBits)) return false;
				make_pairs(t);
			}
			// ---- the symbols of the block.  While at least 8 bytes of input and a longest match plus a copy's overshoot of
			// output are left, nothing in the loop looks at the ends of the buffers; the last stretch runs with every check.
			bool done = false;
			while (!done && (size_t)(b.end - b.ip) >= 8 && (size_t)(oend - op) >= 258 + 3 + 8) {
				{   // refill, the branch-free form
					uint64_t w;
					memcpy(&w, b.ip, 8);
					b.buf |= w << b.cnt;
					b.ip += (63 - b.cnt) >> 3;
					b.cnt |= 56;
				}
				uint32_t e = tb->pair[b.buf & ((1u << kLitBits) - 1)];
				if (e & kLiteral) {
					// one or two literals per lookup, up to four lookups (44 of the 56 bits) per refill; two bytes are stored either
					// way (the second is overwritten when the entry held one literal: room for it was checked above)
					int k = 0;
					do {
						const uint16_t two = (uint16_t)(e >> 16);
						memcpy(op, &two, 2);
						op += 1 + ((e >> 8) & 1u);
						b.buf >>= (uint8_t)e; b.cnt -= (int)(uint8_t)e;
						e = tb->pair[b.buf & ((1u << kLitBits) - 1)];
					} while ((e & kLiteral) && ++k < 4);
					continue;
				}
				if (e & kSub) { b.drop(kLitBits); e = tb->lit[(e >> 16) + b.peek((int)(e & 0xffu))]; }
				b.drop((int)(e & 0xffu));
				if (e & kLiteral) { *op++ = (uint8_t)(e >> 16); continue; }
				if (e & (kEob | kBad)) {
					if (e & kBad) return false;
					done = true;
					break;
				}
				const uint32_t length = (e >> 16) + b.take((int)((e >> 8) & 0xfu));
				if (b.cnt < 32) b.refill();
				uint32_t d = tb->dist[b.peek(kDistBits)];
				if (d & kSub) { b.drop(kDistBits); d = tb->dist[(d >> 16) + b.peek((int)(d & 0xffu))]; }
				if (d & kBad) return false;
				b.drop((int)(d & 0xffu));
				const uint32_t dist = (d >> 16) + b.take((int)((d >> 8) & 0xfu));
				if (dist > (size_t)(op - out)) return false;
				const uint8_t *src = op - dist;
				uint8_t *dst = op;
				op += length;
				if (dist >= 8) {
					do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < op);
				} else if (dist == 1) {
					memset(dst, *src, length);                            // a run
				} else {
					do { *dst++ = *src++; } while (dst < op);
				}
			}
			while (!done) {
				b.refill();                                               // >= 56 bits: a length with its extra bits, a distance with its, and change
				uint32_t e = tb->lit[b.peek(kLitBits)];
				if (e & kSub) { b.drop(kLitBits); e = tb->lit[(e >> 16) + b.peek((int)(e & 0xffu))]; }
				b.drop((int)(e & 0xffu));
				if (e & kLiteral) {
					if (op == oend) return false;
					*op++ = (uint8_t)(e >> 16);
					continue;
				}
				if (e & (kEob | kBad)) {
					if (e & kBad) return false;
					break;
				}
				const uint32_t length = (e >> 16) + b.take((int)((e >> 8) & 0xfu));
				if (b.cnt < 32) b.refill();
				uint32_t d = tb->dist[b.peek(kDistBits)];
				if (d & kSub) { b.drop(kDistBits); d = tb->dist[(d >> 16) + b.peek((int)(d & 0xffu))]; }
				if (d & kBad) return false;
				b.drop((int)(d & 0xffu));
				const uint32_t dist = (d >> 16) + b.take((int)((d >> 8) & 0xfu));
				if (dist > (size_t)(op - out) || length > (size_t)(oend - op)) return false;
				const uint8_t *src = op - dist;
				for (uint32_t i = 0; i < length; i++) op[i] = src[i];
				op += length;
			}
			if (b.past_end()) return false;
		} else {
			return false;
		}
		if (bfinal) break;
	}
	return op == oend && !b.past_end();
}

// ---- CRC-32 (the polynomial of gzip) by carry-less multiplication: 64 bytes folded per step --------------------------
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc, const uint8_t *buf, size_t len)
{
	// len is a multiple of 16 and at least 64; crc is the running value as zlib holds it (already inverted by the caller)
	alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
	alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
	alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
	alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
	x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
	x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
	x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
	x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
	x0 = _mm_load_si128((const __m128i *)k1k2);
	buf += 64; len -= 64;
	while (len >= 64) {
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
		y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
		buf += 64; len -= 64;
	}
	x0 = _mm_load_si128((const __m128i *)k3k4);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {
		x2 = _mm_loadu_si128((const __m128i *)buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16; len -= 16;
	}
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_loadl_epi64((const __m128i *)k5k0);
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_load_si128((const __m128i *)poly);
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// crc32 of zlib's kind (crc = 0 to start, the running value to continue); tail_crc is zlib's own crc32 for the bytes the
// folded part leaves over, handed in so that this file needs no zlib header
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len, uint32_t (*tail_crc)(uint32_t, const uint8_t *, size_t))
{
#if defined(__x86_64__)
	static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (ok && len >= 64) {
		const size_t n = len & ~(size_t)15;
		crc = ~crc32_clmul(~crc, buf, n);
		buf += n; len -= n;
	}
#endif
	return len ? tail_crc(crc, buf, len) : crc;
}

}  // namespace host
