// Differential harness for the host-side text helpers (host_common.cpp) — the Python test compares every answer with the
// oracle's independent implementation.  stdin: one hex-encoded byte string per line; stdout: utf8_valid trim_end_len
// trim_start_off is_ascii bc_start bc_end (bc_* = -1 when the ` BC:` field is absent; text functions run only on valid UTF-8).
#include <cstdio>
#include <iostream>
#include <string>

#include "host_common.h"

int main()
{
	std::string hex;
	while (std::getline(std::cin, hex)) {
		std::string s;
		for (size_t i = 0; i + 1 < hex.size(); i += 2) s.push_back((char)std::stoi(hex.substr(i, 2), nullptr, 16));
		const bool ok = host::utf8_valid(reinterpret_cast<const uint8_t *>(s.data()), s.size());
		long te = -1, ts = -1, b0 = -1, b1 = -1;
		if (ok) {
			te = (long)host::trim_end_len(s);
			ts = (long)host::trim_start_off(s);
			if ((long)host::trim_end_len(s.data(), s.size()) != te) te = -2;       // the char* overload must agree
		}
		size_t st, en;
		if (host::find_bc_field(s, st, en)) { b0 = (long)st; b1 = (long)en; }
		printf("%d %ld %ld %d %ld %ld\n", ok ? 1 : 0, te, ts, host::is_ascii(s) ? 1 : 0, b0, b1);
	}
	return 0;
}
