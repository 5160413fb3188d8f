"""`from taichi.tools.video import VideoManager` (scripts/async/async_mpm.py:4,48-49): only the frame directory is used."""
import os


class VideoManager:
    def __init__(self, output_dir, width=0, height=0):
        self.directory, self.width, self.height = output_dir, width, height
        self.frame_directory = os.path.join(output_dir, "frames")
        os.makedirs(self.frame_directory, exist_ok=True)

    def get_frame_directory(self):
        return self.frame_directory

    def write_frame(self, img):
        return None

    def make_video(self):  # rendering is outside the scope of this build: the frames are .bgeo files
        return None
