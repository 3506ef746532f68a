// Unit test of host::GzWriter (block-parallel gzip): several files written interleaved, contents checked by the
// Python side after decompression.  usage: gz_writer_test <dir> <nfiles> <total_bytes_per_file>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "host_common.h"

int main(int argc, char **argv)
{
	if (argc != 4) return 2;
	const std::string dir = argv[1];
	const int nf = atoi(argv[2]);
	const size_t total = (size_t)atoll(argv[3]);
	std::vector<std::unique_ptr<host::GzWriter>> w;
	for (int i = 0; i < nf; i++) w.emplace_back(new host::GzWriter(dir + "/f" + std::to_string(i) + ".gz"));
	std::vector<size_t> done(nf, 0);
	uint64_t x = 88172645463325252ull;
	bool more = true;
	while (more) {
		more = false;
		for (int i = 0; i < nf; i++) {
			if (done[i] >= total) continue;
			more = true;
			x ^= x << 13; x ^= x >> 7; x ^= x << 17;
			size_t n = 1 + (size_t)(x % 9000);
			if (n > total - done[i]) n = total - done[i];
			std::string s(n, 'x');
			for (size_t k = 0; k < n; k++) s[k] = (char)('A' + ((done[i] + k) * (i + 3)) % 23);
			w[i]->write(s);
			done[i] += n;
		}
	}
	for (auto &p : w) p->close();
	return 0;
}
