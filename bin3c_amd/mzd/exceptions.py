"""
Application exception family of the reference (mzd/exceptions.py:2-52): same class names and messages,
so ``except ApplicationException`` blocks and log text written against bin3C keep working.
"""


class ApplicationException(Exception):
    """Root of every error the application reports and exits on (bin3C.py:213-216)."""

    def __init__(self, message):
        Exception.__init__(self, message)
        self.message = message


def _fixed(name, doc, fmt):
    """Build an ApplicationException subclass whose message is ``fmt`` filled with the constructor arguments."""

    def __init__(self, *args):
        ApplicationException.__init__(self, fmt.format(*args))

    return type(name, (ApplicationException,), {'__init__': __init__, '__doc__': doc})


UnknownEnzymeException = type('UnknownEnzymeException', (ApplicationException,), {
    '__doc__': 'An enzyme name that the restriction database does not know',
    '__init__': lambda self, target, similar: ApplicationException.__init__(
        self, '{} is undefined, but its similar to: {}'.format(target, ', '.join(similar)))})
UnknownOrientationStateException = _fixed('UnknownOrientationStateException', 'Orientation other than +1/-1',
                                          'unknown orientation state [{}].')
NoneAcceptedException = _fixed('NoneAcceptedException', 'All sequences were excluded during filtering',
                               'all sequences were excluded')
TooFewException = _fixed('TooFewException', 'Method requires a minimum of nodes',
                         'More than {} sequences are required to apply {}')
NoRemainingClustersException = _fixed('NoRemainingClustersException', 'No cluster survived a filter', '{}')
NoReportException = _fixed('NoReportException', 'Clustering does not contain a report',
                           'No clusters contained an ordering')
ZeroLengthException = _fixed('ZeroLengthException', 'Sequence of zero length', 'Sequence [{}] has zero length')
ParsingError = _fixed('ParsingError', 'An error during input parsing', '{}')
