"""ffmpeg stdin pipe (reference src/ffmpeg_pipe.rs:21-153): spawns
`ffmpeg -hide_banner -loglevel error -stats -f image2pipe -framerate 60 -y -i - [-i <wav>] <args> <out>`, writes PNG
bytes to stdin, repeats the previous frame for dropped/missing frames, respawns when the timestamp changes and checks
the exit status when closed."""
import os
import shutil
import subprocess
import sys

from . import _ffi
from .compressor import generate_timestamp


class FfmpegPipe:
    def __init__(self, audio_directory, output_directory, output_filename, ffmpeg_args):
        self.audio_directory = audio_directory
        self.output_directory = output_directory
        self.output_filename = output_filename
        self.ffmpeg_args = list(ffmpeg_args)
        self.child = None
        self.timestamp = None
        self.prev_bytes = None

    @staticmethod
    def available():
        return shutil.which("ffmpeg") is not None

    def write(self, video_frame, png_bytes, timestamp=None):
        if not png_bytes and self.prev_bytes is None:
            return
        if self.child is None or self.timestamp_has_changed(timestamp):
            self.re_spawn_process(timestamp)
        stdin = self.child.stdin
        if not png_bytes:
            print(f"Warning: Frame {video_frame.frame_number} is {video_frame.status}. Duplicating previous frame to "
                  "maintain a steady frame rate.", file=sys.stderr)
            stdin.write(self.prev_bytes)
        else:
            stdin.write(png_bytes)
            self.prev_bytes = png_bytes

    def timestamp_has_changed(self, timestamp):
        if timestamp == self.timestamp:
            return False
        if self.output_filename is not None:
            print("Warning: Compressed data contains multiple videos but only writing one file: "
                  f"{self.output_filename}", file=sys.stderr)
        return True

    def command_line(self, timestamp):
        """The argv the reference builds (src/ffmpeg_pipe.rs:70-100)."""
        argv = ["ffmpeg", "-hide_banner", "-loglevel", "error", "-stats", "-f", "image2pipe", "-framerate", "60",
                "-y", "-i", "-"]
        output_filename, output_path = self.output_filename_and_path(timestamp)
        wav = self.look_for_wav_file(output_filename)
        if wav is not None:
            argv += ["-i", wav]
        return argv + self.ffmpeg_args + [output_path]

    def re_spawn_process(self, timestamp):
        self.close()
        self.timestamp = timestamp
        self.child = subprocess.Popen(self.command_line(timestamp), stdin=subprocess.PIPE, stdout=subprocess.PIPE)

    def output_filename_and_path(self, timestamp=None):
        directory = self.output_directory if self.output_directory is not None else "."
        filename = self.output_filename
        if filename is None:
            ts = timestamp if timestamp is not None else self.timestamp
            filename = f"{generate_timestamp(ts)}.mp4"
        return filename, os.path.join(directory, filename)

    def look_for_wav_file(self, output_filename):
        if self.audio_directory is None:
            return None
        path = os.path.splitext(os.path.join(self.audio_directory, output_filename))[0] + ".wav"
        if os.path.exists(path):
            return path
        print(f"Skipping audio because {path} doesn't exist", file=sys.stderr)
        return None

    def close(self):
        """The reference's Drop: waits for ffmpeg and panics on a non-zero exit status."""
        child, self.child = self.child, None
        if child is None:
            return
        try:
            child.stdin.close()
        except BrokenPipeError:
            pass
        status = child.wait()
        if status != 0:
            raise _ffi.RendererError(f"ffmpeg exited with {status}")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
