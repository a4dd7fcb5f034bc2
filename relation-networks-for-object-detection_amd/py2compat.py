"""Import hook for the reference's Python-2 sources: modules found in a registered directory whose only Python-3
problem is the `print` STATEMENT are compiled after an in-memory `lib2to3` `fix_print` pass (a mechanical
statement -> function rewrite); nothing is written to disk and no reference file is modified.
(`cPickle`, `xrange` and the removed numpy aliases are handled by `mx.install(py2_shims=True)`.)"""
import importlib.abc
import importlib.util
import os
import sys

_DIRS = []


class _Loader(importlib.abc.SourceLoader):
    def __init__(self, fullname, path):
        self.fullname, self.path = fullname, path

    def get_filename(self, fullname):
        return self.path

    def get_data(self, path):
        with open(path, 'rb') as f:
            return f.read()

    def source_to_code(self, data, path, *, _optimize=-1):
        src = data.decode('utf-8') if isinstance(data, bytes) else data
        try:
            return compile(src, path, 'exec', dont_inherit=True, optimize=_optimize)
        except SyntaxError:
            from lib2to3.refactor import RefactoringTool
            src3 = str(RefactoringTool(['lib2to3.fixes.fix_print']).refactor_string(src if src.endswith('\n') else src + '\n', path))
            return compile(src3, path, 'exec', dont_inherit=True, optimize=_optimize)

    def get_code(self, fullname):                                  # no bytecode cache next to the reference's files
        return self.source_to_code(self.get_data(self.path), self.path)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if '.' in fullname:
            return None
        for d in _DIRS:
            f = os.path.join(d, fullname + '.py')
            if os.path.isfile(f):
                return importlib.util.spec_from_loader(fullname, _Loader(fullname, f), origin=f)
        return None


_finder = _Finder()


def add_source_dir(path):
    """Top-level modules of `path` are imported through the print-statement tolerant loader (takes precedence over
    sys.path for those names)."""
    path = os.path.abspath(path)
    if path not in _DIRS:
        _DIRS.append(path)
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
