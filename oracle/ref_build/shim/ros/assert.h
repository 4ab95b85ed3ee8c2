#pragma once
#include "ros.h"
