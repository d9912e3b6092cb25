"""Command-line flag registry with the ``tf.app.flags`` / absl surface the reference CLIs use.

The reference defines its flags through ``tf.app.flags`` (helper/args.py:13-98) and its CLIs add
their own (sr.py:34, evaluate.py:38-39) before ``tf.app.run()`` parses ``sys.argv``.  absl is not
part of this stack, so the small subset of its behaviour those scripts rely on is provided here:

* ``DEFINE_integer / DEFINE_float / DEFINE_boolean (DEFINE_bool) / DEFINE_string``
* ``--name=value``, ``--name value``, single or double dash; booleans also as ``--name`` /
  ``--noname`` / ``--name=true|false|1|0|yes|no|t|f``
* ``FLAGS.name`` attribute access (defaults are visible before parsing), assignment, ``--help``
* unknown flags are fatal, positional leftovers are handed to ``main`` (evaluate.py:44-47)
"""

import sys


class FlagError(Exception):
    pass


class _Flag:
    __slots__ = ("name", "default", "help", "kind", "value", "present")

    def __init__(self, name, default, help_text, kind):
        self.name = name
        self.default = default
        self.help = help_text
        self.kind = kind
        self.value = default
        self.present = False

    def parse(self, text):
        if self.kind == "boolean":
            low = text.strip().lower()
            if low in ("true", "t", "1", "yes", "y"):
                return True
            if low in ("false", "f", "0", "no", "n"):
                return False
            raise FlagError("flag --%s=%s: not a boolean" % (self.name, text))
        if self.kind == "integer":
            try:
                return int(text, 0)
            except ValueError:
                raise FlagError("flag --%s=%s: invalid literal for an integer" % (self.name, text))
        if self.kind == "float":
            try:
                return float(text)
            except ValueError:
                raise FlagError("flag --%s=%s: invalid literal for a float" % (self.name, text))
        return text


class FlagValues:
    def __init__(self):
        object.__setattr__(self, "_flags", {})
        object.__setattr__(self, "_parsed", False)

    # -- registration ------------------------------------------------------------------------
    def _define(self, name, default, help_text, kind):
        flags = object.__getattribute__(self, "_flags")
        if name in flags:
            old = flags[name]
            if old.kind == kind and old.default == default:
                return          # re-import of a CLI module: harmless
            raise FlagError("The flag '%s' is defined twice." % name)
        flags[name] = _Flag(name, default, help_text, kind)

    # -- access ------------------------------------------------------------------------------
    def __getattr__(self, name):
        flags = object.__getattribute__(self, "_flags")
        if name in flags:
            return flags[name].value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        flags = object.__getattribute__(self, "_flags")
        if name not in flags:
            raise AttributeError("unknown flag '%s'" % name)
        flags[name].value = value

    def __contains__(self, name):
        return name in object.__getattribute__(self, "_flags")

    def __iter__(self):
        return iter(object.__getattribute__(self, "_flags"))

    def flag_values_dict(self):
        return {n: f.value for n, f in object.__getattribute__(self, "_flags").items()}

    def is_parsed(self):
        return object.__getattribute__(self, "_parsed")

    def reset(self):
        """Back to defaults (used by tests that parse several command lines in one process)."""
        for f in object.__getattribute__(self, "_flags").values():
            f.value = f.default
            f.present = False
        object.__setattr__(self, "_parsed", False)

    # -- parsing -----------------------------------------------------------------------------
    def __call__(self, argv):
        """Parse ``argv`` (argv[0] is the program name); returns [argv[0]] + positional leftovers."""
        flags = object.__getattribute__(self, "_flags")
        rest = [argv[0]] if argv else [""]
        i = 1
        while i < len(argv):
            arg = argv[i]
            i += 1
            if arg == "--":
                rest.extend(argv[i:])
                break
            if not arg.startswith("-") or arg == "-":
                rest.append(arg)
                continue
            body = arg.lstrip("-")
            if body in ("help", "helpfull", "h", "helpshort"):
                print(self.usage(argv[0]))
                sys.exit(0)
            name, eq, text = body.partition("=")
            flag = flags.get(name)
            if flag is None and not eq and name.startswith("no") and name[2:] in flags \
                    and flags[name[2:]].kind == "boolean":
                flag = flags[name[2:]]
                flag.value, flag.present = False, True
                continue
            if flag is None:
                raise FlagError("Unknown command line flag '%s'" % name)
            if flag.kind == "boolean" and not eq:
                flag.value, flag.present = True, True
                continue
            if not eq:
                if i >= len(argv):
                    raise FlagError("Missing value for flag --%s" % name)
                text = argv[i]
                i += 1
            flag.value, flag.present = flag.parse(text), True
        object.__setattr__(self, "_parsed", True)
        return rest

    def usage(self, prog=""):
        lines = ["flags for %s:" % prog]
        for name in sorted(object.__getattribute__(self, "_flags")):
            f = object.__getattribute__(self, "_flags")[name]
            lines.append("  --%s: %s\n    (default: %r)" % (("[no]" + name) if f.kind == "boolean" else name, f.help, f.default))
        return "\n".join(lines)


FLAGS = FlagValues()


def DEFINE_integer(name, default, help, flag_values=FLAGS):   # noqa: A002 (absl signature)
    flag_values._define(name, default, help, "integer")


def DEFINE_float(name, default, help, flag_values=FLAGS):     # noqa: A002
    flag_values._define(name, None if default is None else float(default), help, "float")


def DEFINE_boolean(name, default, help, flag_values=FLAGS):   # noqa: A002
    flag_values._define(name, default, help, "boolean")


DEFINE_bool = DEFINE_boolean


def DEFINE_string(name, default, help, flag_values=FLAGS):    # noqa: A002
    flag_values._define(name, default, help, "string")


def run(main, argv=None):
    """``tf.app.run()``: parse the command line, call ``main(leftover_argv)``, exit with its result."""
    argv = list(sys.argv if argv is None else argv)
    try:
        rest = FLAGS(argv)
    except FlagError as exc:
        print("FATAL Flags parsing error: %s\nPass --helpshort or --helpfull to see help on flags." % exc,
              file=sys.stderr)
        sys.exit(1)
    sys.exit(main(rest))
