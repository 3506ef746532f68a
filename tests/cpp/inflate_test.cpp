// host::inflate_raw and host::crc32_fast (seqkit_amd/csrc/host_inflate.cpp) against zlib, under ASan + UBSan:
//  * buffers of many shapes (text-like, runs, random, BAM-like) deflated by zlib at every level and strategy — stored,
//    fixed and dynamic blocks, long and short distances — must inflate to the same bytes;
//  * damaged streams (flipped bits, truncations, wrong sizes) must never crash, and whenever inflate_raw accepts one, zlib
//    accepts it too with the same output (when it gives up, the caller asks zlib: no verdict of its own);
//  * the CRC of random ranges at random alignments equals zlib's.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "host_common.h"

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy)
{
	z_stream zs;
	memset(&zs, 0, sizeof zs);
	deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
	std::vector<uint8_t> out(deflateBound(&zs, (uLong)in.size()) + 64);
	zs.next_in = const_cast<uint8_t *>(in.data()); zs.avail_in = (uInt)in.size();
	zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
	deflate(&zs, Z_FINISH);
	out.resize(zs.total_out);
	deflateEnd(&zs);
	return out;
}

static bool zlib_inflate(const uint8_t *in, size_t n, std::vector<uint8_t> &out, size_t want)
{
	z_stream zs;
	memset(&zs, 0, sizeof zs);
	inflateInit2(&zs, -15);
	out.assign(want + 1, 0);
	zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n;
	zs.next_out = out.data(); zs.avail_out = (uInt)want;
	const int rc = inflate(&zs, Z_FINISH);
	const bool ok = rc == Z_STREAM_END && zs.avail_out == 0;
	inflateEnd(&zs);
	out.resize(want);
	return ok;
}

int main(int argc, char **argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 300;
	std::mt19937_64 rng(99);
	auto pick = [&](size_t n) { return (size_t)(rng() % n); };
	size_t ok_count = 0, damaged = 0, accepted_damaged = 0;
	for (int it = 0; it < rounds; it++) {
		const size_t n = it < 8 ? (size_t)it : (pick(4) == 0 ? pick(300) : pick(66000));
		std::vector<uint8_t> src(n);
		switch (pick(6)) {
		case 0: for (auto &c : src) c = (uint8_t)rng(); break;                                   // incompressible
		case 1: for (auto &c : src) c = (uint8_t)"ACGT"[pick(4)]; break;                         // bases
		case 2: { uint8_t v = 0; for (auto &c : src) { if (pick(50) == 0) v = (uint8_t)rng(); c = v; } break; }      // runs (distance 1)
		case 3: for (size_t i = 0; i < n; i++) src[i] = (uint8_t)(i < 300 ? rng() : src[i - 1 - pick(299)]); break;    // short back-references
		case 4: for (size_t i = 0; i < n; i++) src[i] = (uint8_t)(33 + (pick(10) ? 30 + pick(11) : pick(40))); break;     // quality-like
		default: for (size_t i = 0; i < n; i++) src[i] = (uint8_t)(i % 7 == 0 ? rng() : "the quick brown fox "[i % 20]); break;
		}
		const int levels[] = {0, 1, 3, 6, 9};
		const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
		const std::vector<uint8_t> comp = deflate_raw(src, levels[pick(5)], strategies[pick(5)]);
		std::vector<uint8_t> out(n + 1, 0xAB);
		if (!host::inflate_raw(comp.data(), comp.size(), out.data(), n) || (n && memcmp(out.data(), src.data(), n) != 0) || out[n] != 0xAB) {
			fprintf(stderr, "MISMATCH on a valid stream: round %d, %zu bytes\n", it, n);
			return 1;
		}
		ok_count++;
		// the right stream with the wrong size must be refused
		if (n > 0 && host::inflate_raw(comp.data(), comp.size(), out.data(), n - 1)) { fprintf(stderr, "accepted a short output buffer (round %d)\n", it); return 1; }
		{
			std::vector<uint8_t> big(n + 2);
			if (host::inflate_raw(comp.data(), comp.size(), big.data(), n + 1)) { fprintf(stderr, "accepted a long output buffer (round %d)\n", it); return 1; }
		}
		// damage
		for (int d = 0; d < 6 && !comp.empty(); d++) {
			std::vector<uint8_t> bad(comp);
			if (d < 4) bad[pick(bad.size())] ^= (uint8_t)(1u << pick(8));
			else bad.resize(pick(bad.size()));
			// the buffer handed in is exactly as long as the stream: ASan sees any read past it
			std::vector<uint8_t> exact(bad.begin(), bad.end());
			std::vector<uint8_t> o2(n + 1, 0xCD), ref;
			const bool mine = host::inflate_raw(exact.data(), exact.size(), o2.data(), n);
			damaged++;
			if (o2[n] != 0xCD) { fprintf(stderr, "wrote past the output buffer\n"); return 1; }
			if (mine) {
				accepted_damaged++;
				if (!zlib_inflate(exact.data(), exact.size(), ref, n) || (n && memcmp(ref.data(), o2.data(), n) != 0)) {
					fprintf(stderr, "accepted a damaged stream that zlib refuses or decodes differently (round %d, damage %d)\n", it, d);
					return 1;
				}
			}
		}
	}
	// CRC
	std::vector<uint8_t> buf(1 << 20);
	for (auto &c : buf) c = (uint8_t)rng();
	auto zcrc = [](uint32_t c, const uint8_t *p, size_t n) { return (uint32_t)crc32(c, p, (uInt)n); };
	for (int it = 0; it < 3000; it++) {
		const size_t off = pick(4096), len = pick(it % 3 == 0 ? 200 : 70000);
		const uint32_t start = it % 4 == 0 ? 0u : (uint32_t)rng();
		if (host::crc32_fast(start, buf.data() + off, len, zcrc) != (uint32_t)crc32(start, buf.data() + off, (uInt)len)) {
			fprintf(stderr, "CRC mismatch: off %zu len %zu start %08x\n", off, len, start);
			return 1;
		}
	}
	printf("ok: %zu streams, %zu damaged (%zu of them still valid for both decoders)\n", ok_count, damaged, accepted_damaged);
	return 0;
}
