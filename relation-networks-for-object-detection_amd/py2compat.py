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
            return compile(fix_print_statements(src, path), path, 'exec', dont_inherit=True, optimize=_optimize)

    def get_code(self, fullname):                                  # no bytecode cache next to the reference's files
        return self.source_to_code(self.get_data(self.path), self.path)


def _fix_print_lines(src):
    """Minimal Python-2 print-statement rewriter (`print a, b`, `print >>f, a`, trailing comma, bare `print`, arguments
    continued over several lines inside brackets): the fallback when lib2to3 is gone (removed from the standard library in
    Python 3.13)."""
    import re

    def depth(text):
        d, q = 0, None
        for ch in text:
            if q:
                q = None if ch == q else q
            elif ch in '"\'':
                q = ch
            elif ch == '#':
                break
            elif ch in '([{':
                d += 1
            elif ch in ')]}':
                d -= 1
        return d

    lines, out, i = src.split('\n'), [], 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r'^(\s*)print(?![\w(=.\[])\s*(.*?)\s*$', line)
        if not m:
            out.append(line); i += 1
            continue
        indent, rest = m.group(1), m.group(2)
        while depth(rest) > 0 and i + 1 < len(lines):          # the argument list continues on the next line(s)
            i += 1
            rest += '\n' + lines[i].rstrip()
        comment = ''
        if '\n' not in rest and '#' in rest and rest.count('"') % 2 == 0 and rest.count("'") % 2 == 0:
            rest, comment = rest[:rest.index('#')].rstrip(), '  ' + rest[rest.index('#'):]
        kw = ''
        if rest.startswith('>>'):
            dest, _, rest = rest[2:].partition(',')
            kw = ', file=%s' % dest.strip()
            rest = rest.strip()
        if rest.endswith(','):
            rest, kw = rest[:-1].rstrip(), kw + ", end=' '"
        args = rest + kw if rest else kw.lstrip(', ')
        out.append('%sprint(%s)%s' % (indent, args, comment))
        i += 1
    return '\n'.join(out)


def fix_print_statements(src, path='<source>'):
    """Python-2 `print` statements -> calls, in memory: lib2to3's fix_print when the interpreter still ships it, else the
    single-line rewriter above (enough for the reference's files: their only Python-3 problem is the print statement)."""
    src = src if src.endswith('\n') else src + '\n'
    try:
        from lib2to3.refactor import RefactoringTool
    except ImportError:
        return _fix_print_lines(src)
    return str(RefactoringTool(['lib2to3.fixes.fix_print']).refactor_string(src, path))


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
