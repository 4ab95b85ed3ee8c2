// TEST INFRASTRUCTURE: empty stand-in (the factor sources include it through parameters.h but use nothing from it).
#pragma once
