// TEST INFRASTRUCTURE: stand-in for the ROS logging/assert macros the Cerberus factor sources reference.
#pragma once
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <map>
#include <queue>
#include <set>
#include <string>
#include <vector>
#define ROS_INFO(...) ((void)0)
#define ROS_DEBUG(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_DEBUG_STREAM(x) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
#define ROS_ASSERT(c) assert(c)
#define ROS_ASSERT_MSG(c, ...) assert(c)
#define ROS_BREAK() std::abort()
namespace ros {
inline bool ok() { return true; }
}
