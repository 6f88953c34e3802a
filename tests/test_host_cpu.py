"""CPU tests of the host-side mirror and of the C-ABI boundary (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "renderer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", text)) - {"rh_map_callback"})


def test_library_exports_every_declared_symbol():
    from renderer_amd import _ffi
    from renderer_amd import build
    build.build_library()
    lib = C.CDLL(_ffi.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/renderer_hip.h but not exported"
    # the ctypes binding covers exactly the declared surface
    assert sorted(_ffi.SIGNATURES) == declared
    assert lib.rh_abi_version() == 23


def test_shader_code_objects_are_gfx950_elf():
    from renderer_amd import Compiler
    import renderer_amd
    d = os.path.join(os.path.dirname(renderer_amd.__file__), "shaders")
    built = Compiler.compile_shaders(d)
    assert len(built) >= 4
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    for path in built:
        assert open(path, "rb").read(4) == b"\x7fELF"
        if os.path.exists(readelf):
            out = subprocess.check_output([readelf, "-h", "-s", path]).decode()
            assert "gfx950" in out and "rh_module_info" in out
            assert ("rh_bin_count" in out) == path.endswith(".vert.co")
            assert ("rh_tile_1" in out) == path.endswith(".frag.co")


def test_buffer_growth_rule():
    from renderer_amd.buffer import HEADROOM, INITIAL_SIZE, next_power_of_two
    assert INITIAL_SIZE == 64 and HEADROOM == 1024
    assert next_power_of_two(8_000_000 + HEADROOM) == 8_388_608          # 1 M vec2 instances
    assert next_power_of_two(160_000_000 + HEADROOM) == 268_435_456       # 10 M vec4 instances
    assert next_power_of_two(65 + HEADROOM) == 2048 and next_power_of_two(1024) == 1024 and next_power_of_two(1) == 1


class _Fake:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def test_slot_numbering():
    from renderer_amd.renderer import texture_index, uniform_index
    from renderer_amd import RendererError
    with_sampler, without = _Fake(sampler=object()), _Fake(sampler=None)
    program = _Fake(instances=[1], uniforms=[], textures=[(with_sampler, None)])
    # quads example: slot 0 = i_offset, slot 1 = t_texture, slot 2 = s_sampler
    assert texture_index(1, program) == 0
    with pytest.raises(RendererError, match="sampler is in that slot"):
        texture_index(2, program)
    with pytest.raises(RendererError, match="nothing is in that slot"):
        texture_index(3, program)
    program = _Fake(instances=[1, 2], uniforms=[3], textures=[(with_sampler, None), (without, None), (with_sampler, None)])
    # slots: 0,1 instanced; 2 uniform; 3 tex A; 4 sampler A; 5 tex B; 6 tex C; 7 sampler C
    assert uniform_index(2, program) == 0
    assert [texture_index(i, program) for i in (3, 5, 6)] == [0, 1, 2]
    with pytest.raises(RendererError, match="sampler is in that slot"):
        texture_index(4, program)


def test_format_table():
    from renderer_amd import Format
    assert [int(f) for f in Format] == [0, 1, 2, 3, 4]
    assert [f.bytes_per_texel() for f in Format] == [1, 4, 4, 8, 16]
    assert Format.default() is Format.RgbaU8


def test_capture_layout_rule():
    from renderer_amd.video_recorder import capture_layout
    # quads example: 1:1 viewport in 1920x1080 -> 1080 px rows of 4320 B padded to 4352 B
    assert capture_layout(1080, 1080, 4) == (4320, 4352, 4_700_160)
    # 4K rows are already 256-byte aligned
    assert capture_layout(3840, 2160, 4) == (15360, 15360, 33_177_600)
    assert capture_layout(1, 1, 4) == (4, 256, 256)


def test_video_frame_bincode_known_answers():
    from renderer_amd.video_frame import AtomicUsize, FrameStatus, ImageData, VideoFrame
    from renderer_amd import Format
    # Captured 4K frame: status 0, Some, 3840 (fb 00 0f), 2160 (fb 70 08), RgbaU8 = 2, 15360 (fb 00 3c) twice,
    # frame 7, 33177600 = 0x01fa4000 (fc 00 40 fa 01), budget 66355200 = 0x03f48000 (fc 00 80 f4 03)
    f = VideoFrame(FrameStatus.Captured, ImageData.Bytes(b""), 3840, 2160, Format.RgbaU8, 15360, 15360, 7, 33177600,
                   AtomicUsize(66355200))
    f._dropped = True
    enc = f.encode()
    assert enc.hex() == "0001" "fb000f" "fb7008" "02" "fb003c" "fb003c" "07" "fc0040fa01" "fc0080f403"
    g, used = VideoFrame.decode(enc)
    assert used == len(enc) and g.status is FrameStatus.Captured and g.image_data is not None and g.image_data.bytes() == b""
    assert (g.width, g.height, g.format, g.frame_number, g.frame_size_in_bytes) == (3840, 2160, Format.RgbaU8, 7, 33177600)
    # Dropped frame: no image data, Option tag 0
    d = VideoFrame(FrameStatus.Dropped, None, 250, 251, Format.RgbaU8, 1000, 1024, 300, 257024, AtomicUsize(0))
    assert d.encode().hex() == "0100" "fa" "fbfb00" "02" "fbe803" "fb0004" "fb2c01" "fc00ec0300" "00"
    g, _ = VideoFrame.decode(d.encode())
    assert g.status is FrameStatus.Dropped and g.image_data is None and g.width == 250 and g.height == 251


def test_recording_position_adds_rgba8_target():
    from renderer_amd.pipeline import Pipeline, RecordingPosition
    from renderer_amd import BlendMode, Format, Target
    p = Pipeline.__new__(Pipeline)
    p.blend_mode = BlendMode.pre_multiplied_alpha(); p.targets = [Target.Screen()]
    p.position_in_recording = RecordingPosition.NONE
    states = p.create_color_target_states()
    assert [(s.format, s.src_factor, s.dst_factor) for s in states] == [(int(Format.BgraU8), 1, 3)]
    p.position_in_recording = RecordingPosition.NotLast
    states = p.create_color_target_states()
    assert [s.format for s in states] == [int(Format.BgraU8), int(Format.RgbaU8)]
    p.handle = None


def test_cpp_host_mirror_builds():
    """The C++ mirror of the reference API compiles against the C ABI header and links with the library."""
    import subprocess
    from renderer_amd import build
    exes = build.build_examples()
    assert [os.path.basename(e) for e in exes] == ["quads_main_thread", "quads_render_thread"]
    for exe in exes:
        assert os.path.exists(exe)
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 2 and "usage" in out.stderr


def test_bench_capture_plan_shrinks_with_ranks_and_room(monkeypatch):
    """bench.py sizes its recorded-frames leg from the available memory (the stream files go to /dev/shm when there is one,
    and a round's bytes are pinned twice) and the free space where the files go: the largest plan when there is room,
    smaller ones as more ranks share the host, never below the smallest."""
    import collections
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    Usage = collections.namedtuple("Usage", "total used free")
    import psutil
    Mem = collections.namedtuple("Mem", "available")
    root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    in_memory = root == "/dev/shm"
    monkeypatch.setattr(psutil, "virtual_memory", lambda: Mem(available=300e9))
    monkeypatch.setattr(shutil, "disk_usage", lambda path: Usage(300e9, 0, 300e9))
    plans = [bench.capture_plan(n, None, 0) for n in (1, 2, 4, 8)]
    assert all(p[3] == root for p in plans)
    if in_memory:
        # one rank: 4096 frames of 30 MB in memory + two pinned rounds of 1024 = 200 GB of the 210 GB it may take
        assert [p[:3] for p in plans] == [(4096, 1024, 1), (1024, 512, 1), (512, 256, 2), (128, 128, 1)]
    else:
        assert [p[:3] for p in plans] == [(4096, 1024, 1), (2048, 1024, 1), (1024, 512, 1), (512, 256, 2)]
    monkeypatch.setattr(shutil, "disk_usage", lambda path: Usage(252e9, 251e9, 1e9))
    assert bench.capture_plan(1, None, 0)[:3] == (128, 128, 1)


def test_option_numbers_of_the_python_mirror_match_the_header():
    """renderer_amd/_ffi.py repeats the RH_OPTION_* values of include/renderer_hip.h."""
    from renderer_amd import _ffi
    header = open(os.path.join(ROOT, "include", "renderer_hip.h")).read()
    declared = {name: int(value) for name, value in re.findall(r"RH_OPTION_([A-Z_0-9]+) = (\d+)", header)}
    assert len(declared) >= 8
    for name, value in declared.items():
        assert getattr(_ffi, "OPTION_" + name) == value, name


def test_bench_refuses_a_gpu_count_the_launcher_did_not_start():
    """bench.py never scales a value by its --gpus argument: with a launcher's WORLD_SIZE that disagrees it leaves with
    an error before touching a GPU, and started plainly with --gpus N on a host with fewer GPUs it refuses too."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="1")
    out = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "must agree" in out.stderr and out.stdout.strip() == ""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "RH_REHEARSE_ON_ONE_GPU")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, bench, "--gpus", "64"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "GPU(s)" in out.stderr and out.stdout.strip() == ""
