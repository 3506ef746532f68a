"""CPU: the host side of the neighbourhood table (seqkit_amd/csrc/sk_lut.cpp) — built for hundreds of random sheets
(wildcard columns, separators, duplicates, seven letters, every length to 20) under ASan + UBSan, every observed barcode
looked up with a C++ model of the kernel's arithmetic and compared with the reference's loop written out plainly
(src/fasta_demultiplex.rs:154-194).  tests/cpp/lut_test.cpp holds both."""
import os
import subprocess


def test_neighbour_table_build_and_lookup_model(tmp_path):
    from seqkit_amd import build
    exe = tmp_path / "lut_test"
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-o", str(exe), os.path.join(build.REPO, "tests", "cpp", "lut_test.cpp"), os.path.join(build.CSRC, "sk_lut.cpp")], check=True)
    out = subprocess.run([str(exe), "400"], stdout=subprocess.PIPE, check=True, env={"ASAN_OPTIONS": "detect_leaks=0"}).stdout.decode()
    assert out.startswith("ok:"), out
    built = int(out.split()[1])
    assert built > 100, out
