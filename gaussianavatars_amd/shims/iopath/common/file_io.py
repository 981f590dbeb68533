"""PathManager subset used by the reference's vendored OBJ loader (utils/pytorch3d_load_obj.py:47): local files only."""
import os


class PathManager:
    def open(self, path, mode="r", **kwargs):
        return open(path, mode)

    def exists(self, path):
        return os.path.exists(path)

    def isfile(self, path):
        return os.path.isfile(path)

    def get_local_path(self, path, **kwargs):
        return str(path)


g_pathmgr = PathManager()
