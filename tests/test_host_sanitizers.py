"""CPU: the host-only native code (packer, Powell state machine) under AddressSanitizer + UBSan."""
import os
import subprocess


def test_host_code_under_asan_ubsan(tmp_path, repo_root):
    exe = tmp_path / "host_sanity"
    src = [os.path.join(repo_root, "tests", "native", "host_sanity.cpp"), os.path.join(repo_root, "nlml_hpe_amd", "csrc", "pack.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-o", str(exe)] + src
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host sanity ok" in res.stdout
