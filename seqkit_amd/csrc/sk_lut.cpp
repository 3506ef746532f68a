// sk_lut.cpp — host side of the neighbourhood table (sk_lut.h): enumerate the keys of a sheet, decide each with the
// reference's loop (src/fasta_demultiplex.rs:154-194, distance :269-277), place them.  Plain C++: compiled into the
// library by hipcc and into tests/cpp/lut_test.cpp by g++.
#include "sk_lut.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

namespace sk {

namespace {
inline bool is_wildcard(uint8_t b) { return b == 'N' || b == 'U'; }      // src/fasta_demultiplex.rs:273

struct Key { uint8_t cls[kLutMaxLen]; uint32_t A, B; };

void pack_classes(const uint8_t *cls, uint32_t &A, uint32_t &B, int wide)
{
	uint32_t c[5] = {0, 0, 0, 0, 0};
	for (int k = 0; k < (wide ? 8 : kLutMaxLen); k++) c[k >> 2] |= (uint32_t)cls[k] << (8 * (k & 3));
	lut_pack(c, A, B, wide);
}
}  // namespace

constexpr size_t kMaxKeys = 220000;      // what two tables of 2^18 slots hold at 42 %

// two-choice cuckoo placement of n keys whose mixed word under seed sd is xof(q, sd): table 1 is indexed by the top nb bits,
// table 2 by the next nb (lut_slot, lut_side2); at most 42 % full, nb in [nb_min, nb_max].  where[slot] = key or -1.
template <typename XOf> bool cuckoo_place(size_t n, int nb_min, int nb_max, XOf xof, int &nb_out, uint32_t &seed_out, std::vector<int> &where)
{
	int nb = nb_min;
	while (nb < nb_max && ((size_t)1 << nb) * 84 < n * 100) nb++;
	for (; nb <= nb_max; nb++) {
		const size_t nslots = (size_t)1 << nb;
		for (uint32_t tr = 1; tr <= 32; tr++) {
			const uint32_t sd = tr * 0x9E3779B9u;
			where.assign(2 * nslots, -1);
			bool placed = true;
			for (size_t q = 0; q < n && placed; q++) {
				int cur = (int)q, side = 0, kicks = 0;
				for (;;) {
					const uint32_t x = xof((size_t)cur, sd);
					const size_t at = side == 0 ? lut_slot(x, nb) : nslots + lut_slot(lut_side2(x, nb), nb);
					std::swap(cur, where[at]);
					if (cur < 0) break;
					side ^= 1;                                                  // the evicted key goes to its slot in the other table
					if (++kicks > 500) { placed = false; break; }
				}
			}
			if (placed) { nb_out = nb; seed_out = sd; return true; }
		}
	}
	return false;
}

struct SheetShape {                      // what both forms need to know about a sheet
	int S, L, max_diff, sep, W1, W2, sh, other;
	int wide = 0, sh2 = 0;               // wide classes (sk_lut.h): 4 bits, the top one g[(b >> sh2) & 7]
	uint8_t g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tab2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	bool counting[kLutMaxLen];
	uint8_t tab[8];
	std::vector<uint8_t> alts;
	std::vector<Key> rows;               // the rows in class space, key positions
	int key_pos(int k) const { return sep < 0 || k < sep ? k : 4 * W1 + (k - sep - 1); }
};

static void fill_common(const SheetShape &sh, const uint8_t *sheet, LutDev &d)
{
	d = LutDev{};
	d.W1 = sh.W1; d.W2 = sh.W2;
	d.sh = sh.sh;
	d.tab_lo = (uint32_t)sh.tab[0] | ((uint32_t)sh.tab[1] << 8) | ((uint32_t)sh.tab[2] << 16) | ((uint32_t)sh.tab[3] << 24);
	d.tab_hi = (uint32_t)sh.tab[4] | ((uint32_t)sh.tab[5] << 8) | ((uint32_t)sh.tab[6] << 16) | ((uint32_t)sh.tab[7] << 24);
	d.other = (uint32_t)sh.other * 0x01010101u;
	d.wide = sh.wide; d.sh2 = sh.sh2;
	auto word = [](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
	d.g_lo = word(sh.g); d.g_hi = word(sh.g + 4);
	d.tab2_lo = word(sh.tab2); d.tab2_hi = word(sh.tab2 + 4);
	uint8_t keep[kLutMaxLen] = {0};
	for (int k = 0; k < sh.L; k++) if (sh.counting[k] && k != sh.sep) keep[sh.key_pos(k)] = sh.wide ? 15 : 7;
	pack_classes(keep, d.keepA, d.keepB, sh.wide);
	d.sep_off = sh.sep;
	d.sep_val = sh.sep < 0 ? 0u : (uint32_t)sheet[sh.sep];
	d.max_diff = sh.max_diff;
	d.idx_shift = 24;
	d.idx_mask = 0x7fu;
}

// The factored form (sk_lut.h): one table per half of an `i7+i5` sheet, one of the sheet's pairs.  false: some half key lies
// within distance 1 of two different half-barcodes (the halves are not all >= 3 apart), or a table does not fit.
static bool build_pair(const SheetShape &sh, const uint8_t *sheet, LutHost &out, int lds_budget)
{
	if (sh.sep < 0 || sh.W1 != sh.W2 || sh.W1 > 2) return false;
	const int hw = 4 * sh.W1;                                                   // key positions per half
	auto half_word = [&](const uint8_t *cls) {
		uint32_t c[2] = {0, 0};
		for (int k = 0; k < hw; k++) c[k >> 2] |= (uint32_t)cls[k] << (8 * (k & 3));
		return sh.W1 == 1 ? c[0] : lut_pack_half(c[0], c[1], sh.wide);
	};
	struct HalfKey { uint32_t A; int h, d; };
	std::vector<HalfKey> hk[2];
	std::vector<int> half_of[2];                                                // sample -> half id
	int nh[2] = {0, 0};
	for (int side = 0; side < 2; side++) {
		std::unordered_map<uint32_t, int> id_of;                                // a half-barcode's word -> its id
		std::vector<const uint8_t *> first_row;
		half_of[side].resize((size_t)sh.S);
		for (int s = 0; s < sh.S; s++) {
			const uint8_t *cls = sh.rows[(size_t)s].cls + side * hw;
			auto it = id_of.find(half_word(cls));
			if (it == id_of.end()) { it = id_of.emplace(half_word(cls), (int)first_row.size()).first; first_row.push_back(cls); }
			half_of[side][(size_t)s] = it->second;
		}
		nh[side] = (int)first_row.size();
		if (nh[side] > 1024) return false;                                      // a half id has 10 bits
		std::unordered_map<uint32_t, int> owner;                                // a half KEY's word -> the one half-barcode within distance 1 of it
		auto add = [&](uint32_t A, int h, int d) {
			auto ins = owner.emplace(A, h);
			if (!ins.second) return ins.first->second == h;                         // another half-barcode's key as well: the factoring would not be the loop
			hk[side].push_back({A, h, d});
			return true;
		};
		for (int h = 0; h < nh[side]; h++) {
			uint8_t v[8];
			memcpy(v, first_row[(size_t)h], (size_t)hw);
			if (!add(half_word(v), h, 0)) return false;
			if (sh.max_diff < 1) continue;
			for (int k = 0; k < sh.L; k++) {
				if (!sh.counting[k] || k == sh.sep) continue;
				const int kp = sh.key_pos(k) - side * hw;
				if (kp < 0 || kp >= hw) continue;
				const uint8_t own = v[kp];
				for (uint8_t alt : sh.alts) {
					if (alt == own) continue;
					v[kp] = alt;
					if (!add(half_word(v), h, 1)) return false;
				}
				v[kp] = own;
			}
		}
	}
	struct PairKey { uint32_t pk; int first, last; };
	std::vector<PairKey> pairs;
	{
		std::unordered_map<uint32_t, size_t> at;
		for (int s = 0; s < sh.S; s++) {
			const uint32_t pk = (uint32_t)half_of[0][(size_t)s] | ((uint32_t)half_of[1][(size_t)s] << 10);
			auto it = at.find(pk);
			if (it == at.end()) { at.emplace(pk, pairs.size()); pairs.push_back({pk, s, s}); }
			else pairs[it->second].last = s;                                        // rows in sheet order: the last one seen is the last argmin
		}
	}
	int nb[3];
	uint32_t seed[3];
	std::vector<int> where[3];
	for (int tb = 0; tb < 3; tb++) {
		const size_t n = tb < 2 ? hk[tb].size() : pairs.size();
		auto xof = [&](size_t q, uint32_t sd) { return lut_mix(tb < 2 ? hk[tb][q].A : pairs[q].pk, 0u, sd); };
		if (!cuckoo_place(n, 4, 14, xof, nb[tb], seed[tb], where[tb])) return false;
	}
	// What a free slot holds.  With 4-bit classes eight columns fill the word, so EVERY 32-bit value is some observed half
	// (0xFFFFFFFF = eight bytes of class 15: a poly-G index read on a mixed-case sheet): the free word is a value that is no KEY of
	// its table, and a free slot's second word carries a distance no max_diff admits — an observed half that equals the free word
	// "hits" the free slot and is refused by `tot <= max_diff` like any other miss, at no instruction in the kernel.  (A pair key
	// has 20 bits: 0xFFFFFFFF is none.)
	uint32_t free_word[3] = {kLutPairFree, kLutPairFree, kLutPairFree};
	for (int tb = 0; tb < 2; tb++) {
		std::unordered_set<uint32_t> keys;
		for (const HalfKey &k : hk[tb]) keys.insert(k.A);
		while (keys.count(free_word[tb])) free_word[tb]--;
	}
	const size_t n1 = (size_t)2 << nb[0], n2 = (size_t)2 << nb[1], np = (size_t)2 << nb[2];
	const size_t bytes = (n1 + n2 + np) * 8;
	if (bytes > (size_t)lds_budget) return false;
	out.slots.assign((n1 + n2 + np) * 2, 0u);
	for (int tb = 0; tb < 3; tb++) {
		const size_t base = tb == 0 ? 0 : (tb == 1 ? n1 : n1 + n2);
		for (size_t i = 0; i < where[tb].size(); i++) {
			uint32_t *e = &out.slots[2 * (base + i)];
			const int q = where[tb][i];
			if (q < 0) { e[0] = free_word[tb]; e[1] = tb < 2 ? kLutPairFreeVal : 0u; }
			else if (tb < 2) { e[0] = hk[tb][(size_t)q].A; e[1] = (uint32_t)hk[tb][(size_t)q].h | ((uint32_t)hk[tb][(size_t)q].d << 16); }
			else { e[0] = pairs[(size_t)q].pk; e[1] = (uint32_t)pairs[(size_t)q].first | ((uint32_t)pairs[(size_t)q].last << 16); }
		}
	}
	out.amb.clear();
	out.n_keys = hk[0].size() + hk[1].size() + pairs.size();
	fill_common(sh, sheet, out.dev);
	LutPairDev &p = out.dev.pair;
	p.nb1 = nb[0]; p.nb2 = nb[1]; p.nbp = nb[2];
	p.off2 = (uint32_t)n1; p.offp = (uint32_t)(n1 + n2);
	p.seed1 = seed[0]; p.seed2 = seed[1]; p.seedp = seed[2];
	{	// the halves' keep masks: the counting columns' class bits in each half's word
		uint8_t keep[2][8] = {{0}, {0}};
		for (int k = 0; k < sh.L; k++)
			if (sh.counting[k] && k != sh.sep) { const int kp = sh.key_pos(k); keep[kp / hw][kp % hw] = sh.wide ? 15 : 7; }
		p.keep1 = half_word(keep[0]);
		p.keep2 = half_word(keep[1]);
	}
	p.bytes = (int)bytes;
	return true;
}

bool lut_build(const uint8_t *sheet, int S, int L, int max_diff, LutHost &out, int lds_budget)
{
	if (!sheet || S < 1 || S > kLutMaxSamples || L < 1 || L > kLutMaxLen || max_diff < 0 || max_diff > 1) return false;
	SheetShape shp;
	shp.S = S; shp.L = L; shp.max_diff = max_diff;
	// columns: counting (part of the key) or ignored (every row has a wildcard there).  A row that has a wildcard in a counting
	// column matches any byte there: it is enumerated once per class of that column (below), so a few such columns per row
	// are affordable and many are not (the bound further down refuses the sheet then)
	bool (&counting)[kLutMaxLen] = shp.counting;
	for (int k = 0; k < kLutMaxLen; k++) counting[k] = false;
	bool mixed = false;
	for (int k = 0; k < L; k++) {
		int nw = 0;
		for (int s = 0; s < S; s++) nw += is_wildcard(sheet[(size_t)s * L + k]) ? 1 : 0;
		counting[k] = nw != S;
		mixed = mixed || (nw != 0 && nw != S);
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
	shp.sep = sep; shp.W1 = W1; shp.W2 = W2;
	auto key_pos = [&](int k) { return shp.key_pos(k); };                       // column -> position in the key
	// the letters of the key columns and a 3-bit function of a byte that separates them
	bool is_letter[256] = {false};
	std::vector<uint8_t> letters;
	for (int k = 0; k < L; k++) {
		if (!counting[k] || k == sep) continue;
		for (int s = 0; s < S; s++) {
			const uint8_t b = sheet[(size_t)s * L + k];
			if (!is_wildcard(b) && !is_letter[b]) { is_letter[b] = true; letters.push_back(b); }
		}
	}
	if (letters.size() > 15) return false;
	int sh = -1;
	for (int t = 0; t <= 5 && sh < 0 && letters.size() <= 7; t++) {
		bool used[8] = {false}, ok = true;
		for (uint8_t b : letters) {
			const int i = (b >> t) & 7;
			if (used[i]) { ok = false; break; }
			used[i] = true;
		}
		if (ok) sh = t;
	}
	if (sh < 0) {
		// Wide classes (sk_lut.h): a 3-bit window that leaves at most two letters per index, and a function g of another window
		// that tells the two apart wherever there are two.  Only shapes whose key is one word per table: at most 8 columns
		// without a separator, or two halves of at most 8 beside one (the factored form; no row with wildcards of its own then)
		if (sep < 0 ? W1 > 2 : (W1 > 2 || mixed)) return false;
		for (int t = 0; t <= 5 && sh < 0; t++) {
			for (int t2 = 0; t2 <= 5 && sh < 0; t2++) {
				for (int gm = 0; gm < 256 && sh < 0; gm++) {
					bool used[16] = {false}, ok = true;
					for (uint8_t b : letters) {
						const int i = ((b >> t) & 7) | (((gm >> ((b >> t2) & 7)) & 1) << 3);
						if (used[i]) { ok = false; break; }
						used[i] = true;
					}
					if (ok) { sh = t; shp.wide = 1; shp.sh2 = t2; for (int j = 0; j < 8; j++) shp.g[j] = (uint8_t)(((gm >> j) & 1) << 3); }
				}
			}
		}
		if (sh < 0) return false;
	}
	shp.sh = sh;
	const int n_classes = shp.wide ? 16 : 8;
	auto index_of = [&](uint8_t b) { return ((b >> sh) & 7) | (shp.wide ? shp.g[(b >> shp.sh2) & 7] : 0); };
	uint8_t (&tab)[8] = shp.tab;
	bool used[16] = {false};
	for (int i = 0; i < 8; i++) tab[i] = shp.tab2[i] = (uint8_t)(((i ^ 1) & 7) << sh);      // a byte with ANOTHER index: nothing with index i equals it
	for (uint8_t b : letters) { (index_of(b) < 8 ? tab : shp.tab2)[index_of(b) & 7] = b; used[index_of(b)] = true; }
	int other = 0;
	while (other < n_classes && used[other]) other++;                         // <= 7 (15) letters: one of the 8 (16) is free
	if (other >= n_classes) return false;
	shp.other = other;
	std::vector<uint8_t> &alts = shp.alts;
	for (uint8_t b : letters) alts.push_back((uint8_t)index_of(b));
	alts.push_back((uint8_t)other);

	// rows in class space
	std::vector<Key> &rows = shp.rows;
	rows.resize((size_t)S);
	int n_key_cols = 0;
	for (int k = 0; k < L; k++) n_key_cols += (counting[k] && k != sep) ? 1 : 0;
	std::vector<std::vector<int>> wild((size_t)S);                              // per row: the key positions where it has a wildcard
	size_t bound = 0;                                                           // keys the enumeration below will produce, at most
	for (int s = 0; s < S; s++) {
		memset(&rows[(size_t)s], 0, sizeof(Key));
		for (int k = 0; k < L; k++) {
			if (!counting[k] || k == sep) continue;
			const uint8_t b = sheet[(size_t)s * L + k];
			if (is_wildcard(b)) wild[(size_t)s].push_back(key_pos(k));
			else rows[(size_t)s].cls[key_pos(k)] = (uint8_t)index_of(b);
		}
		pack_classes(rows[(size_t)s].cls, rows[(size_t)s].A, rows[(size_t)s].B, shp.wide);
		size_t variants = 1;
		for (size_t j = 0; j < wild[(size_t)s].size() && variants <= kMaxKeys; j++) variants *= alts.size();
		const size_t fixed = (size_t)n_key_cols - wild[(size_t)s].size();
		bound += variants * (1 + (max_diff ? fixed * (alts.size() - 1) : 0));
		if (bound > kMaxKeys) return false;
	}
	// A full-key table that will not fit the workgroup's LDS: a sheet with a separator is looked up half by half when that is
	// exact and those tables fit (its exactness argument is about rows without wildcards in key columns)
	if (!mixed) {
		int nbe = kLutMinBits;
		while (((size_t)1 << nbe) * 84 < bound * 100) nbe++;
		if (lds_budget > 0 && (shp.wide || ((size_t)16 << nbe) > (size_t)lds_budget) && build_pair(shp, sheet, out, lds_budget)) return true;
	}
	if (shp.wide && sep >= 0) return false;                                     // (wide classes with a separator: the factored form or none)
	// Enumerate every row and every row with one key column changed, and DECIDE on the way (src/fasta_demultiplex.rs:154-166:
	// lowest distance, first and last row attaining it, rows in sheet order).  The rows within distance 1 of a key are exactly
	// the rows that generate it here — it is that row, or one substitution away from it — so a key's decision is the minimum
	// over its generators: no loop over all rows per key (that loop was 1 s for 1 000 samples x 20 columns).
	struct Decision { int diff, first, last; };
	std::vector<Key> keys;
	std::vector<Decision> dec;
	{
		std::unordered_map<uint64_t, size_t> at;
		auto add = [&](Key k, int s, int d) {
			pack_classes(k.cls, k.A, k.B, shp.wide);
			auto ins = at.emplace(((uint64_t)k.A << 32) | k.B, keys.size());
			if (ins.second) { keys.push_back(k); dec.push_back({d, s, s}); return; }
			Decision &e = dec[ins.first->second];
			if (d < e.diff) e = {d, s, s};                                      // rows come in ascending order: the first to attain a distance is the first argmin
			else if (d == e.diff) e.last = s;
		};
		for (int s = 0; s < S; s++) {
			const std::vector<int> &ws = wild[(size_t)s];
			bool is_wild[kLutMaxLen] = {false};
			for (int kp : ws) is_wild[kp] = true;
			std::vector<size_t> odo(ws.size(), 0);                               // one variant of the row per assignment of classes to its wildcards
			for (;;) {
				Key base = rows[(size_t)s];
				for (size_t j = 0; j < ws.size(); j++) base.cls[ws[j]] = alts[odo[j]];
				add(base, s, 0);
				for (int k = 0; k < L && max_diff >= 1; k++) {
					if (!counting[k] || k == sep || is_wild[key_pos(k)]) continue;
					Key v = base;
					for (uint8_t alt : alts) {
						if (alt == base.cls[key_pos(k)]) continue;
						v.cls[key_pos(k)] = alt;
						add(v, s, 1);
					}
				}
				size_t j = 0;
				while (j < odo.size() && ++odo[j] == alts.size()) odo[j++] = 0;
				if (j == odo.size()) break;
			}
		}
	}
	const int idx_bits = S <= 128 ? 7 : 10;
	std::vector<int16_t> amb;
	std::vector<int> idx(keys.size());
	{
		std::unordered_map<uint32_t, int> pair_at;
		for (size_t q = 0; q < keys.size(); q++) {
			if (dec[q].first == dec[q].last) { idx[q] = dec[q].first; continue; }
			const uint32_t pr = ((uint32_t)dec[q].first << 16) | (uint32_t)dec[q].last;
			auto it = pair_at.find(pr);
			if (it == pair_at.end()) {
				if (amb.size() / 2 >= ((size_t)1 << idx_bits)) return false;
				it = pair_at.emplace(pr, (int)(amb.size() / 2)).first;
				amb.push_back((int16_t)dec[q].first); amb.push_back((int16_t)dec[q].last);
			}
			idx[q] = it->second;
		}
	}
	// two-choice cuckoo, at most 42 % full; the tag (33 - nb bits) must leave room for idx and the flag
	int nb = 0;
	uint32_t seed = 0;
	std::vector<int> where;
	auto xof = [&](size_t q, uint32_t sd) { return lut_mix(keys[q].A, keys[q].B, sd); };
	if (!cuckoo_place(keys.size(), idx_bits + 2, 18, xof, nb, seed, where)) return false;
	const size_t nslots = (size_t)1 << nb;
	const int idx_shift = 31 - idx_bits;
	out.slots.assign(2 * nslots * 2, 0u);
	for (size_t i = 0; i < 2 * nslots; i++) {
		uint32_t *e = &out.slots[2 * i];
		// the tag is one bit wider than what the slot index leaves: its top bit is the slot's lowest bit, so a free slot holds
		// the tag no key of this slot can have and the kernels need no "is it free" test beside the tag compare
		const uint32_t tag_mask = (uint32_t)(((uint64_t)1 << (33 - nb)) - 1);
		if (where[i] < 0) { e[0] = kLutFree; e[1] = (uint32_t)(~i & 1u) << (32 - nb); continue; }
		const size_t q = (size_t)where[i];
		const uint32_t x = lut_mix(keys[q].A, keys[q].B, seed);
		const uint32_t tag = (i < nslots ? x : lut_side2(x, nb)) & tag_mask;
		const bool ambiguous = dec[q].first != dec[q].last;
		e[0] = keys[q].B | ((uint32_t)dec[q].diff << 31);
		e[1] = tag | ((uint32_t)idx[q] << idx_shift) | (ambiguous ? 0x80000000u : 0u);
	}
	out.amb = amb;
	out.n_keys = keys.size();
	LutDev &d = out.dev;
	fill_common(shp, sheet, d);
	d.nb = nb; d.mask = (int)(nslots - 1);
	d.seed = seed; d.tag_mask = (uint32_t)(((uint64_t)1 << (33 - nb)) - 1);
	d.idx_shift = idx_shift;
	d.idx_mask = (1u << idx_bits) - 1u;
	return true;
}

}  // namespace sk
