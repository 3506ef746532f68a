// sk_lut.cpp — host side of the neighbourhood table (sk_lut.h): enumerate the keys of a sheet, decide each with the
// reference's loop (src/fasta_demultiplex.rs:154-194, distance :269-277), place them.  Plain C++: compiled into the
// library by hipcc and into tests/cpp/lut_test.cpp by g++.
#include "sk_lut.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>

namespace sk {

namespace {
inline bool is_wildcard(uint8_t b) { return b == 'N' || b == 'U'; }      // src/fasta_demultiplex.rs:273
inline uint32_t rotr32(uint32_t x, int r) { return r ? (x >> r) | (x << (32 - r)) : x; }

struct Key { uint8_t cls[kLutMaxLen]; uint32_t A, B; };

void pack_classes(const uint8_t *cls, uint32_t &A, uint32_t &B)
{
	uint32_t c[5] = {0, 0, 0, 0, 0};
	for (int k = 0; k < kLutMaxLen; k++) c[k >> 2] |= (uint32_t)cls[k] << (8 * (k & 3));
	lut_pack(c, A, B);
}
}  // namespace

bool lut_build(const uint8_t *sheet, int S, int L, int max_diff, LutHost &out)
{
	if (!sheet || S < 1 || S > kLutMaxSamples || L < 1 || L > kLutMaxLen || max_diff < 0 || max_diff > 1) return false;
	// columns: counting (no row has a wildcard there) or ignored (every row has one); a wildcard in some rows only would
	// make the key depend on the row
	bool counting[kLutMaxLen] = {false};
	for (int k = 0; k < L; k++) {
		int nw = 0;
		for (int s = 0; s < S; s++) nw += is_wildcard(sheet[(size_t)s * L + k]) ? 1 : 0;
		if (nw != 0 && nw != S) return false;
		counting[k] = nw == 0;
	}
	// the separator: a counting column with one letter in every row, a letter no other counting column uses, that cuts the
	// row into two segments of equally many dwords (1 + 1 or 2 + 2: `i7+i5`); the kernel reads the segments on their own,
	// so the key has no hole where the separator was
	int sep = -1;
	for (int k = 1; k + 1 < L && sep < 0; k++) {
		if (!counting[k]) continue;
		const int w1 = (k + 3) / 4, w2 = (L - k - 1 + 3) / 4;
		if (w1 != w2 || w1 > 2) continue;
		const uint8_t x = sheet[k];
		bool ok = true;
		for (int s = 1; s < S && ok; s++) ok = sheet[(size_t)s * L + k] == x;
		for (int j = 0; j < L && ok; j++) {
			if (j == k || !counting[j]) continue;
			for (int s = 0; s < S && ok; s++) ok = sheet[(size_t)s * L + j] != x;
		}
		if (ok) sep = k;
	}
	const int W1 = sep < 0 ? (L + 3) / 4 : (sep + 3) / 4, W2 = sep < 0 ? 0 : W1;
	auto key_pos = [&](int k) { return sep < 0 || k < sep ? k : 4 * W1 + (k - sep - 1); };      // column -> position in the key
	// the letters of the key columns and a 3-bit function of a byte that separates them
	bool is_letter[256] = {false};
	std::vector<uint8_t> letters;
	for (int k = 0; k < L; k++) {
		if (!counting[k] || k == sep) continue;
		for (int s = 0; s < S; s++) {
			const uint8_t b = sheet[(size_t)s * L + k];
			if (!is_letter[b]) { is_letter[b] = true; letters.push_back(b); }
		}
	}
	if (letters.size() > 7) return false;
	int sh = -1;
	for (int t = 0; t <= 5 && sh < 0; t++) {
		bool used[8] = {false}, ok = true;
		for (uint8_t b : letters) {
			const int i = (b >> t) & 7;
			if (used[i]) { ok = false; break; }
			used[i] = true;
		}
		if (ok) sh = t;
	}
	if (sh < 0) return false;
	auto index_of = [&](uint8_t b) { return (b >> sh) & 7; };
	uint8_t tab[8];
	bool used[8] = {false};
	for (int i = 0; i < 8; i++) tab[i] = (uint8_t)(((i ^ 1) & 7) << sh);      // a byte with ANOTHER index: nothing with index i equals it
	for (uint8_t b : letters) { tab[index_of(b)] = b; used[index_of(b)] = true; }
	int other = 0;
	while (used[other]) other++;                                              // <= 7 letters: one of the 8 is free
	std::vector<uint8_t> alts;
	for (uint8_t b : letters) alts.push_back((uint8_t)index_of(b));
	alts.push_back((uint8_t)other);

	// rows in class space; enumerate every row and every row with one key column changed
	std::vector<Key> rows((size_t)S);
	for (int s = 0; s < S; s++) {
		memset(&rows[(size_t)s], 0, sizeof(Key));
		for (int k = 0; k < L; k++)
			if (counting[k] && k != sep) rows[(size_t)s].cls[key_pos(k)] = (uint8_t)index_of(sheet[(size_t)s * L + k]);
		pack_classes(rows[(size_t)s].cls, rows[(size_t)s].A, rows[(size_t)s].B);
	}
	std::vector<Key> keys;
	{
		std::unordered_map<uint64_t, int> seen;
		auto add = [&](Key k) {
			pack_classes(k.cls, k.A, k.B);
			if (seen.emplace(((uint64_t)k.A << 32) | k.B, 0).second) keys.push_back(k);
		};
		for (int s = 0; s < S; s++) {
			add(rows[(size_t)s]);
			if (max_diff < 1) continue;
			for (int k = 0; k < L; k++) {
				if (!counting[k] || k == sep) continue;
				Key v = rows[(size_t)s];
				for (uint8_t alt : alts) {
					if (alt == rows[(size_t)s].cls[key_pos(k)]) continue;
					v.cls[key_pos(k)] = alt;
					add(v);
				}
			}
		}
	}
	// decide: src/fasta_demultiplex.rs:154-166 (first / last argmin over the rows in sheet order)
	struct Decision { int diff, first, last; };
	std::vector<Decision> dec(keys.size());
	for (size_t q = 0; q < keys.size(); q++) {
		int lowest = 0x7fffffff, first = 0, last = 0;
		for (int s = 0; s < S; s++) {
			int d = 0;
			for (int k = 0; k < kLutMaxLen; k++) d += (keys[q].cls[k] != rows[(size_t)s].cls[k]) ? 1 : 0;      // ignored positions are 0 on both sides
			if (d < lowest) { lowest = d; first = s; last = s; }
			else if (d == lowest) last = s;
		}
		if (lowest > max_diff) return false;                                  // cannot happen: every key is a row or one step from one
		dec[q] = {lowest, first, last};
	}
	std::vector<int16_t> amb;
	std::vector<int> idx(keys.size());
	{
		std::unordered_map<uint32_t, int> pair_at;
		for (size_t q = 0; q < keys.size(); q++) {
			if (dec[q].first == dec[q].last) { idx[q] = dec[q].first; continue; }
			const uint32_t pr = ((uint32_t)dec[q].first << 16) | (uint32_t)dec[q].last;
			auto it = pair_at.find(pr);
			if (it == pair_at.end()) {
				if (amb.size() / 2 >= 128) return false;
				it = pair_at.emplace(pr, (int)(amb.size() / 2)).first;
				amb.push_back((int16_t)dec[q].first); amb.push_back((int16_t)dec[q].last);
			}
			idx[q] = it->second;
		}
	}
	// two-choice cuckoo, at most 42 % full
	int nb = kLutMinBits;
	while (((size_t)1 << nb) * 84 < keys.size() * 100) nb++;
	std::vector<int> where;
	uint32_t seed = 0;
	bool placed = false;
	for (; nb <= 16 && !placed; nb++) {
		const size_t nslots = (size_t)1 << nb;
		const uint32_t mask = (uint32_t)(nslots - 1);
		for (uint32_t tr = 1; tr <= 32 && !placed; tr++) {
			const uint32_t sd = tr * 0x9E3779B9u;
			where.assign(2 * nslots, -1);
			placed = true;
			for (size_t q = 0; q < keys.size() && placed; q++) {
				int cur = (int)q, side = 0, kicks = 0;
				for (;;) {
					const uint32_t x = lut_mix(keys[(size_t)cur].A, keys[(size_t)cur].B, sd);
					const size_t at = side == 0 ? (x & mask) : nslots + (rotr32(x, nb) & mask);
					std::swap(cur, where[at]);
					if (cur < 0) break;
					side ^= 1;                                                  // the evicted key goes to its slot in the other table
					if (++kicks > 500) { placed = false; break; }
				}
			}
			if (placed) seed = sd;
		}
		if (placed) break;
	}
	if (!placed) return false;
	const size_t nslots = (size_t)1 << nb;
	out.slots.assign(2 * nslots * 2, 0u);
	for (size_t i = 0; i < 2 * nslots; i++) {
		uint32_t *e = &out.slots[2 * i];
		if (where[i] < 0) { e[0] = kLutFree; e[1] = 0; continue; }
		const size_t q = (size_t)where[i];
		const uint32_t x = lut_mix(keys[q].A, keys[q].B, seed);
		const uint32_t tag = (i < nslots ? x : rotr32(x, nb)) >> nb;
		const bool ambiguous = dec[q].first != dec[q].last;
		e[0] = keys[q].B | ((uint32_t)dec[q].diff << 31);
		e[1] = tag | ((uint32_t)idx[q] << 24) | (ambiguous ? 0x80000000u : 0u);
	}
	out.amb = amb;
	out.n_keys = keys.size();
	LutDev &d = out.dev;
	d = LutDev{};
	d.W1 = W1; d.W2 = W2;
	d.nb = nb; d.mask = (int)(nslots - 1);
	d.seed = seed; d.tag_mask = (uint32_t)(((uint64_t)1 << (32 - nb)) - 1);
	d.sh = sh;
	d.tab_lo = (uint32_t)tab[0] | ((uint32_t)tab[1] << 8) | ((uint32_t)tab[2] << 16) | ((uint32_t)tab[3] << 24);
	d.tab_hi = (uint32_t)tab[4] | ((uint32_t)tab[5] << 8) | ((uint32_t)tab[6] << 16) | ((uint32_t)tab[7] << 24);
	d.other = (uint32_t)other * 0x01010101u;
	uint8_t keep[kLutMaxLen] = {0};
	for (int k = 0; k < L; k++) if (counting[k] && k != sep) keep[key_pos(k)] = 7;
	pack_classes(keep, d.keepA, d.keepB);
	d.sep_off = sep;
	d.sep_val = sep < 0 ? 0u : (uint32_t)sheet[sep];
	d.max_diff = max_diff;
	return true;
}

}  // namespace sk
