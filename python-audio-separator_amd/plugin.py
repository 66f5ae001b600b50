"""Registration of the drop-in plugin classes with the reference's orchestrator."""
_PKG = __name__.rsplit(".", 1)[0]
def install(architectures=("mdx", "mdxc", "demucs", "vr")):
    """Make the reference's orchestrator load this package's plugin classes: ``Separator.load_model`` resolves its
    architecture class with ``importlib.import_module("audio_separator.separator.architectures.<arch>_separator")``
    (separator.py:903-904), so registering these modules under those names is the whole integration -- no line of the
    reference changes.  Call before ``Separator.load_model``; returns the list of names registered."""
    import importlib
    import sys
    done = []
    for a in architectures:
        mod = importlib.import_module(f"{_PKG}.architectures.{a}_separator")
        name = f"audio_separator.separator.architectures.{a}_separator"
        sys.modules[name] = mod
        parent = sys.modules.get("audio_separator.separator.architectures")
        if parent is not None:
            setattr(parent, f"{a}_separator", mod)
        done.append(name)
    return done


def uninstall():
    import sys
    for a in ("mdx", "mdxc", "demucs", "vr"):
        name = f"audio_separator.separator.architectures.{a}_separator"
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "__name__", "").startswith(_PKG + "."):
            del sys.modules[name]

