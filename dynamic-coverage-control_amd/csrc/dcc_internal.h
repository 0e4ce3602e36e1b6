// Internal to libdcc_hip.so (not installed): the error channel shared by every translation unit.
#pragma once
#include <string>

// Records `msg` as the calling thread's last error (returned by dcc_last_error()) and returns `code`.
int dcc_fail(int code, const std::string& msg);
