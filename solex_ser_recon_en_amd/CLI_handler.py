"""Command-line flags of SHG_MAIN (reference CLI_handler.py:10-114): same flags, same
option keys, same parsing order.  One addition: the value of -w / -r may also be given
detached (`-w -10:10:1`), which the reference's parser rejects with int('') errors."""
import re
import sys

flag_dictionnary = {
    'h': 'Help',
    'w': 'shift',
    'd': 'flag_display',
    'x': 'ratio_fixe',
    'f': 'save_fit',
    'c': 'clahe_only',
    'p': 'disk_display',
    's': 'crop_width_square',
    't': 'transversalium',
    'm': 'flip_x',
    'r': 'fixed_width',
}

_HELP = [
    ('h', "'Help', display help menu."),
    ('w', "'a,b,c, ...'  produce images at a, b, c ... pixels."),
    ('w', "'x:y:w'  produce images starting at x, finishing at y, every w pixels."),
    ('d', "'flag_display', display all graphics (False by default)"),
    ('x', "'ratio_fixe', disable ellipse fitting"),
    ('f', "'save_fit', save all fits files (False by default)"),
    ('c', "'clahe_only',  only final clahe image is saved (False by default)"),
    ('p', "'disk_display' turn off black disk with protuberance images (False by default)"),
    ('s', "'crop_square_width', crop the width to equal the height (False by default)"),
    ('t', "'disable transversalium', disable transversalium correction (False by default)"),
    ('m', "'mirror flip', mirror flip in x-direction (False by default)"),
    ('r', "'w'  crop width to a constant no. of pixels."),
]


def usage():
    lines = ["SHG_MAIN.py [-hwdxfcpstmr] [file(s) to treat, * allowed]"]
    lines += ["'%s' : %s" % kv for kv in _HELP]
    return '\n'.join(lines)


def parse_shift(text):
    """'a,b,c' | 'x:y' | 'x:y:w' -> list of ints (reference CLI_handler.py:65-74)."""
    parts = text.split(':')
    if len(parts) == 1:
        return [int(x.strip()) for x in text.split(',')]
    if len(parts) == 2:
        return list(range(int(parts[0].strip()), int(parts[1].strip()) + 1))
    if len(parts) == 3:
        return list(range(int(parts[0].strip()), int(parts[1].strip()) + 1, int(parts[2].strip())))
    print('invalid shift input')
    sys.exit()


def treat_flag_at_cli(options, argument):
    """Apply one '-xyz...' argument to options; returns 'w' / 'r' if that flag still waits for a detached value."""
    options['disk_display'] = True
    body = argument[1:]
    i = 0
    pending = None
    while i < len(body):
        ch = body[i]
        if ch == 'h':
            print(usage())
            sys.exit()
        elif ch == 'w':
            j = i + 1
            while j < len(body) and (body[j].isdigit() or body[j] in ':,-'):
                j += 1
            value = body[i + 1:j]
            i = j                                  # the character that ends the value is the next flag
            if value == '':
                pending = 'w'
            else:
                options['shift'] = parse_shift(value)
        elif ch == 'r':
            j = i + 1
            while j < len(body) and body[j].isdigit():
                j += 1
            value = body[i + 1:j]
            i = j
            if value == '':
                pending = 'r'
            else:
                options['fixed_width'] = int(value)
        elif ch == 't':
            options['transversalium'] = False
            i += 1
        elif ch == 'p':
            options['disk_display'] = False
            i += 1
        elif ch == 'x':
            options['ratio_fixe'] = 1
            i += 1
        else:
            if ch in flag_dictionnary:
                options[flag_dictionnary[ch]] = True
            else:
                print('ERROR !!! At least one argument is not accepted')
                print(usage())
            i += 1
    print('options %s' % (options))
    return pending


def handle_CLI(options, argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    serfiles = []
    pending = None
    for argument in argv:
        if pending == 'w' and re.fullmatch(r'[-\d:,]+', argument) and re.search(r'\d', argument):
            options['shift'] = parse_shift(argument)
            pending = None
        elif pending == 'r' and argument.isdigit():
            options['fixed_width'] = int(argument)
            pending = None
        elif argument.startswith('-'):
            if pending:
                raise ValueError('flag -%s needs a value' % pending)
            pending = treat_flag_at_cli(options, argument)
        else:
            if argument.split('.')[-1].upper() in ('SER', 'AVI'):
                serfiles.append(argument)
            else:
                print(f'WARNING: {argument} was not a valid SER or AVI file name and was ignored. '
                      'Remember to use "-" if you want to input a flag')
    if pending:
        raise ValueError('flag -%s needs a value' % pending)
    print('theses files are going to be processed : ', serfiles)
    return serfiles
