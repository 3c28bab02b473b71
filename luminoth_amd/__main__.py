"""`lumi`-style entry point (reference: luminoth/cli.py:12-32): `python -m luminoth_amd <command> [args]` with the
hosted commands train, predict and eval; the reference's checkpoint / cloud / dataset / server groups are outside
the hot-path scope."""
import sys

COMMANDS = {'train': 'luminoth_amd.train', 'predict': 'luminoth_amd.predict', 'eval': 'luminoth_amd.eval'}
NOT_HOSTED = ('checkpoint', 'cloud', 'dataset', 'server')


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print('usage: python -m luminoth_amd {%s} [options]' % '|'.join(sorted(COMMANDS)))
        return 0 if argv else 2
    cmd, rest = argv[0], argv[1:]
    if cmd in NOT_HOSTED:
        print('`%s` is not hosted (outside the hot-path scope); hosted commands: %s' % (cmd, ', '.join(sorted(COMMANDS))),
              file=sys.stderr)
        return 2
    if cmd not in COMMANDS:
        print('unknown command %r; hosted commands: %s' % (cmd, ', '.join(sorted(COMMANDS))), file=sys.stderr)
        return 2
    import importlib
    rc = importlib.import_module(COMMANDS[cmd]).main(rest)
    return 0 if cmd == 'train' or rc is None else int(rc)      # train.main returns the global step reached


if __name__ == '__main__':
    sys.exit(main())
