from .library import Library, KernelError
from .parse_api import parse_api
from .system import (
    get_array_module,
    execution_context,
    num_devices,
    measure_time,
    to_host,
)
