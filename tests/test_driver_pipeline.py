"""The drivers' three-stage window pipeline (samtools_amd/csrc/driver_pipeline.h) with a fake device stage, under ThreadSanitizer:
output order == submission order, held jobs are submitted twice, wait() sees device results, device errors stop the output; and the same
through the text ring (round 6: pieces of a job's text leave while the job is still on the device, the device stage waits for free pieces)."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_pipeline_orders_output_and_is_race_free(tmp_path):
    exe = str(tmp_path / "pipe_test")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(REPO, "tests", "cpu", "pipe_test.cpp"), os.path.join(REPO, "samtools_amd", "csrc", "host_pinned.cpp"),
           "-o", exe, "-pthread", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_NO_PINNED="1"))
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert b"pipe_test OK" in p.stdout
    assert b"ThreadSanitizer" not in p.stderr
