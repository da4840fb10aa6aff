"""Status codes of include/zigz_hip.h as Python exceptions named after the Zig errors they mirror."""


class ZigzError(Exception):
    """error.<name> of the reference, or a backend-specific failure."""

    def __init__(self, code, name, detail=""):
        super().__init__(f"{name} ({code})" + (f": {detail}" if detail else ""))
        self.code = code
        self.name = name


OK = 0
EMPTY_EVALUATIONS = 1
LENGTH_NOT_POWER_OF_TWO = 2
WRONG_NUMBER_OF_VARIABLES = 3
NO_VARIABLES_TO_FIX = 4
NO_VARIABLES = 5
PROTOCOL_ERROR = 6
EMPTY_VALUES = 7
TOO_MANY_VALUES = 8
INDEX_OUT_OF_BOUNDS = 9
POINT_DIMENSION_MISMATCH = 10
NO_QUERIES = 11
TOO_MANY_QUERIES = 12
MAPPING_LENGTH_MISMATCH = 13
INVALID_MAPPING = 14
QUERY_TABLE_MISMATCH = 15
EMPTY_TRACE = 16
OUT_OF_MEMORY = 17
WRONG_NUMBER_OF_CHALLENGES = 18
NO_DEVICE = 100
HIP_ERROR = 101
NOT_CANONICAL = 102
INVALID_ARGUMENT = 103
BAD_STATE = 104
COMM_ERROR = 105
